"""Where a block's lifetime goes inside fusedFaceCellKernel: the -DQGD_FU_CLOCK=1 build of the library (make -C qgdsolver_amd/csrc
BUILD=/tmp/bclk OUT=ab_libs/libqgd_clock.so EXTRA=-DQGD_FU_CLOCK=1) reads s_memtime at the kernel's phase boundaries and leaves the
differences in the new records of the block's own cells; ONE step, then the fields are read back as tick counts.

    QGD_AMD_LIB=ab_libs/libqgd_clock.so python scripts/fused_phase_clock.py [n=200]

rho: start -> the block's lists are there (round 0) | Ux: -> records loaded and staged in LDS (round 1) | Uy: first barrier + vertex values |
Uz: second barrier + the faces | p: third barrier, flux planes into LDS, fourth barrier | e: the cell's sums and advanceCell.
Waves 0 and 1 of a block own its cells, so these are their clocks (cells 0-63 / 64-127 of a block)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd.synthetic import box_initial_fields  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    mesh = q.PolyMesh.box(n, n, n)
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.1 / n / 1.3))
    U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    assert case.fused_info()["fused"]
    case.sync()
    t0 = time.perf_counter()
    case.step_phase(3)
    case.sync()
    wall = time.perf_counter() - t0
    names = ["lists (round 0)", "records + staging (round 1)", "barrier + vertex values", "barrier + faces", "barriers + flux planes", "cell sums + advance"]
    rho = np.asarray(case.field("rho"), dtype=np.float64)
    Uo = np.asarray(case.field("U"), dtype=np.float64).reshape(-1, 3)
    pp = np.asarray(case.field("p"), dtype=np.float64)
    ee = np.asarray(case.field("e"), dtype=np.float64)
    cols = np.stack([rho, Uo[:, 0], Uo[:, 1], Uo[:, 2], pp, ee], axis=1)
    tot = cols.sum(axis=1)
    out = {"n": n, "cells": int(mesh.nCells), "first_step_wall_ms": round(wall * 1e3, 3), "ticks_total_mean": float(tot.mean()),
           "ticks_total_p10_p50_p90": [float(x) for x in np.percentile(tot, [10, 50, 90])], "phases": {}}
    for i, nm in enumerate(names):
        c = cols[:, i]
        out["phases"][nm] = {"mean_ticks": round(float(c.mean()), 1), "share": round(float(c.mean() / tot.mean()), 4),
                             "p10_p50_p90": [round(float(x), 1) for x in np.percentile(c, [10, 50, 90])]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
