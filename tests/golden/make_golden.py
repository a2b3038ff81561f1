#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference ships no golden vectors (and cannot be built or imported here), so these fixtures pin the ORACLE's
output at the time of writing: later edits to the oracle or to the kernels are checked against them
(tests/test_golden.py).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import qgdsolver_amd as q  # noqa: E402  (mesh generator + option struct only; no device code runs here)
import cases  # noqa: E402
from oracle import OracleCase  # noqa: E402
from util import make_mesh, oracle_mesh_of  # noqa: E402

GOLDEN = {
    # name: (mesh kind, scheme, bc function name, init function name, options, steps)
    "box654_jitter_gvp": ("box654_jitter", "GaussVolPoint", None, "box", dict(deltaT=1e-3, mu=1e-3), 10),
    "box654_tri_gvp": ("box654_tri", "GaussVolPoint", None, "box", dict(deltaT=1e-3), 10),
    "step2d_lsq_qgdflux": ("step2d", "leastSquares", "forward_step", "step", dict(deltaT=5e-4), 10),
    "step2d_gvp_qgdflux": ("step2d", "GaussVolPoint", "forward_step", "step", dict(deltaT=5e-4), 10),
    "box654_poly_gvp_mixed": ("box654_poly", "GaussVolPoint", "mixed_box", "box", dict(deltaT=5e-4, mu=1e-3), 10),
    "plane2d_reduced": ("plane2d", "reduced", "empty_z", "plane", dict(deltaT=1e-3, mu=1e-3), 10),
}
FLUX_FIELDS = ["phiJm", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU", "phiwStar", "tauQGDf"]
STATE_FIELDS = ["rho", "U", "p", "e", "rhoU", "rhoE"]


def init_fields(name, C):
    if name == "box":
        return cases.box_initial_fields(C)
    if name == "plane":
        U, T, _ = cases.box_initial_fields(C)
        r2 = (C[:, 0] - 0.5) ** 2 + (C[:, 1] - 0.4) ** 2
        return U, T, 1.0 + 0.1 * np.exp(-r2 / 0.01)
    U = np.zeros((C.shape[0], 3))
    U[:, 0] = 3.0
    return U, 1.0 + 0.05 * np.sin(2.0 * C[:, 0]) * np.cos(3.0 * C[:, 1]), 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])


def apply_bcs(name, case):
    if name == "forward_step":
        cases.forward_step_bcs(case)
    elif name == "mixed_box":
        case.set_bc(0, U=("fixedValue", (0.3, 0.0, 0.0)), T=("fixedValue", 1.0), p=("zeroGradient", None))
        case.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
        case.set_bc(2, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))
        case.set_bc(3, U=("slip", None), T=("fixedValue", 1.05), p=("qgdFlux", None))
        case.set_bc(4, U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("zeroGradient", None))
        case.set_bc(5, U=("slip", None), T=("zeroGradient", None), p=("zeroGradient", None))
    elif name == "empty_z":
        for patch in (4, 5):
            case.set_bc(patch, U=("none", None), T=("none", None), p=("none", None))


def run_case(case_cls_factory, spec):
    """case_cls_factory(mesh, options) -> object with set_bc/set_fields/updateFluxes/step/field"""
    kind, scheme, bc, init, opt, steps = spec
    mesh = make_mesh(kind)
    options = q.default_options(stencil=scheme, **opt)
    case = case_cls_factory(mesh, options)
    apply_bcs(bc, case)
    U, T, p = init_fields(init, mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    case.updateFluxes()
    out = {"flux0_" + n: case.field(n) for n in FLUX_FIELDS}
    case.step(steps)
    out.update({"state_" + n: case.field(n) for n in STATE_FIELDS})
    return out


def oracle_factory(mesh, options):
    return OracleCase(oracle_mesh_of(mesh), options)


if __name__ == "__main__":
    for name, spec in GOLDEN.items():
        data = run_case(oracle_factory, spec)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print(name, {k: v.shape for k, v in data.items()})
