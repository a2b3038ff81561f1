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


# ---- stateless operators: inputs are stored in the fixture next to the oracle's outputs ------------------------------
OPERATOR_GOLDEN = {
    # name: (mesh kind, scheme)
    "op_box654_poly_gvp": ("box654_poly", "GaussVolPoint"),
    "op_plane2d_jitter_lsq": ("plane2d_jitter", "leastSquares"),
}


def operator_inputs(mesh, seed):
    rng = np.random.default_rng(seed)
    nC, nB, nF = mesh.nCells, mesh.nBoundaryFaces, mesh.nFaces
    d = {
        "in_s": rng.standard_normal(nC), "in_sb": rng.standard_normal(nB),
        "in_v": rng.standard_normal((nC, 3)), "in_vb": rng.standard_normal((nB, 3)),
        "in_t": rng.standard_normal((nC, 9)), "in_tb": rng.standard_normal((nB, 9)),
        "in_T": 1.0 + 0.1 * rng.random(nC), "in_Tb": 1.0 + 0.1 * rng.random(nB),
        "in_rho": 1.0 + 0.1 * rng.random(nC), "in_rhob": 1.0 + 0.1 * rng.random(nB),
        "in_tau": 1e-3 * (1.0 + rng.random(nF)), "in_phi": rng.standard_normal(nF), "in_phiJm": rng.standard_normal(nF),
        "in_Y": rng.random(nC), "in_Yb": rng.random(nB),
        "in_phiu": 1e-2 * rng.standard_normal(nF), "in_phiwo": 1e-3 * rng.standard_normal(nF),
        "in_pb": 1.0 + 0.1 * rng.standard_normal(nB), "in_gb": 0.1 * rng.standard_normal(nB),
    }
    if mesh.nGeometricD < 3:  # no velocity in the empty direction
        d["in_v"][:, 2] = 0.0
        d["in_vb"][:, 2] = 0.0
    return d


P_KINDS = ["fixedValue", "zeroGradient", "qhdFlux", "zeroGradient", "none", "none"]


def run_operators(backend, mesh, scheme, d):
    """backend: object with fvsc(op, cell, bnd), qhd(U, T, rho, tau, p, phi), pressure(...), species(...)"""
    out = {}
    out["grad_s"] = backend.fvsc(scheme, "grad_s", d["in_s"], d["in_sb"])
    out["grad_v"] = backend.fvsc(scheme, "grad_v", d["in_v"], d["in_vb"])
    out["div_v"] = backend.fvsc(scheme, "div_v", d["in_v"], d["in_vb"])
    out["div_t"] = backend.fvsc(scheme, "div_t", d["in_t"], d["in_tb"])
    qhd = backend.qhd(scheme, (d["in_v"], d["in_vb"]), (d["in_T"], d["in_Tb"]), (d["in_rho"], d["in_rhob"]), d["in_tau"], 3e-3,
                      (0.0, -9.81, 0.0), (d["in_s"], d["in_sb"]), d["in_phi"])
    for k, v in qhd.items():
        out["qhd_" + k] = v
    sp = backend.species(scheme, (d["in_Y"], d["in_Yb"]), (d["in_v"], d["in_vb"]), d["in_phiJm"], d["in_phi"], d["in_tau"])
    for k, v in sp.items():
        out["species_" + k] = v
    kinds = P_KINDS if mesh.nGeometricD < 3 else P_KINDS[:4] + ["fixedGradient", "zeroGradient"]
    p, phi, info = backend.pressure(d["in_phiu"], d["in_phiwo"], d["in_tau"], np.ones(mesh.nCells), kinds, d["in_pb"], d["in_gb"])
    out["pEqn_p"], out["pEqn_phi"] = p, phi
    return out


class OracleOps:
    def __init__(self, mesh):
        import oracle as orc
        self.orc, self.mesh, self.om = orc, mesh, oracle_mesh_of(mesh)

    def fvsc(self, scheme, op, cell, bnd):
        rc, out = self.om.fvsc(scheme, op, cell, bnd)
        assert rc == 0
        return out

    def qhd(self, scheme, U, T, rho, tau, beta, g, p, phi):
        return self.orc.qhd_fluxes(self.om, scheme, U, T, rho, tau, beta, g, p=p, phi=phi)

    def species(self, scheme, Y, U, phiJm, phi, tau):
        from qgdsolver_amd import qgdfoam

        def call(*a):
            assert self.orc.species_flux(self.om, *a) == 0
        return qgdfoam.speciesFlux(self, scheme, Y, U, phiJm, phi, tau, call=call)

    def pressure(self, phiu, phiwo, tbr, p0, kinds, pb, gb):
        from qgdsolver_amd import qhdfoam
        return qhdfoam.pEqn(self, phiu, phiwo, tbr, p0, kinds, pb, gb, tolerance=1e-13, maxIter=5000, pRefCell=0, pRefValue=0.0,
                            call=lambda *a: self.orc.qhd_pressure(self.om, *a))


def oracle_factory(mesh, options):
    return OracleCase(oracle_mesh_of(mesh), options)


def make_operator_goldens():
    for name, (kind, scheme) in OPERATOR_GOLDEN.items():
        mesh = make_mesh(kind)
        d = operator_inputs(mesh, sum(map(ord, name)))
        out = run_operators(OracleOps(mesh), mesh, scheme, d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d, **{"out_" + k: v for k, v in out.items()})
        print(name, len(out), "outputs")


if __name__ == "__main__":
    make_operator_goldens()
    for name, spec in GOLDEN.items():
        data = run_case(oracle_factory, spec)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print(name, {k: v.shape for k, v in data.items()})
