"""What a cut costs the fused one-launch step, timed on ONE GPU: one k-slab of the n^3 box as rank `r` of `world` would hold
it -- n x n x n/world owned planes + one ghost plane per cut, the ghost records held fixed (no exchange: the message is 10 MB per cut, 0.07 ms
at xGMI speed; this measures the compute side) -- in the plain phase order (0, 1) and in the boundary-layer-first order (0, 10, 11) that overlaps
the exchange, against the UNCUT box with the same number of cells (edge n / world^(1/3): 400^3 / 8 -> 200^3) in the same process.

    python scripts/shard_slab_step.py [n=400] [world=8] [rank=3] [steps=100]

Prints the blocks of each case (qgd_case_fused_info: count, boundary-layer blocks, mean cells per block, faces computed and cell records staged
per cell) and ms per step."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd.halo import slab_range  # noqa: E402
from qgdsolver_amd.synthetic import box_initial_fields  # noqa: E402


def timed(case, order, steps, warmup=20):
    for _ in range(warmup):
        for ph in order:
            case.step_phase(ph)
    case.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for ph in order:
            case.step_phase(ph)
    case.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def run(tag, mesh, owned, orders, steps, deltaT):
    t0 = time.perf_counter()
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=deltaT))
    U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    setup = time.perf_counter() - t0
    fi = case.fused_info()
    out = {"case": tag, "cells_owned": owned, "setup_s": round(setup, 2), "fused": fi["fused"], "blocks": fi["blocks"], "layer_blocks": fi["layerBlocks"],
           "mean_cells_per_block": round(owned / fi["blocks"], 2) if fi["blocks"] else None,
           "faces_computed_per_cell": round(fi["facesComputed"] / owned, 3), "cell_records_staged_per_cell": round(fi["cellsStaged"] / owned, 3),
           "vertex_values_formed_per_cell": round(fi["verticesFormed"] / owned, 3), "lds_bytes": fi["ldsBytes"], "ms_per_step": {}}
    for name, order in orders:
        case.set_fields(U, T, p)
        ms = timed(case, order, steps)
        out["ms_per_step"][name] = round(ms, 4)
        out.setdefault("Mcell_steps_per_s", {})[name] = round(owned / ms / 1e3, 1)
    i = case.info()
    out["min_rho"] = i["minRho"]
    case.close(); dev.close()
    print(json.dumps(out), flush=True)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    dt = 0.1 / n / 1.3
    lo, hi, k_lo, k_hi = slab_range(n, rank, world)
    print(f"# slab: rank {rank} of {world} of the {n}^3 box: owned planes {lo}..{hi}, held planes {k_lo}..{k_hi}; QGD_FUSED_BRICK={os.environ.get('QGD_FUSED_BRICK')}", flush=True)
    slab = q.PolyMesh.box(n, n, n, k_range=(k_lo, k_hi))
    a = run(f"slab {n}x{n}x{hi - lo} (+{(k_hi - k_lo) - (hi - lo)} ghost planes)", slab, n * n * (hi - lo),
            (("plain (0, 1)", (0, 1)), ("boundary layer first (0, 10, 11)", (0, 10, 11))), steps, dt)
    slab.close()
    m = round((n ** 3 / world) ** (1.0 / 3.0))
    box = q.PolyMesh.box(m, m, m)
    b = run(f"uncut box {m}^3", box, m ** 3, (("one launch (3)", (3,)), ("phases (0, 1)", (0, 1))), steps, 0.1 / m / 1.3)
    box.close()
    ref = b["ms_per_step"]["one launch (3)"] * a["cells_owned"] / b["cells_owned"]
    print(f"# slab / uncut box per cell: plain {a['ms_per_step']['plain (0, 1)'] / ref:.3f}, boundary layer first "
          f"{a['ms_per_step']['boundary layer first (0, 10, 11)'] / ref:.3f}")


if __name__ == "__main__":
    main()
