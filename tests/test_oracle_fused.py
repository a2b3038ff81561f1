"""The fused flux assembly of the CPU restatement (orc_case_step_fused: one vertex pass + one face pass, seven face fields stored)
against the field-at-a-time assembly that mirrors the reference's updateFields.H / updateFluxes.H -- what bench.py reports as
cpu_baseline["fused"] must be the same computation, only organised for speed."""
import numpy as np
import pytest

import qgdsolver_amd as q

import cases
from oracle import OracleCase
from util import make_mesh, oracle_mesh_of


def some_bcs(case):
    case.set_bc(0, U=("fixedValue", (0.1, 0.0, 0.0)), T=("fixedValue", 1.05), p=("zeroGradient", None))
    case.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
    case.set_bc(2, U=("slip", None), T=("zeroGradient", None), p=("zeroGradient", None))
    case.set_bc(3, U=("slip", None), T=("zeroGradient", None), p=("zeroGradient", None))


@pytest.mark.parametrize("kind,bc_fn", [("box654_jitter", some_bcs), ("box654", None)])
def test_fused_assembly_is_the_unfused_one(kind, bc_fn):
    mesh = make_mesh(kind)
    fields = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    opt = q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3)
    a, b = OracleCase(oracle_mesh_of(mesh), opt), OracleCase(oracle_mesh_of(mesh), opt)
    for c in (a, b):
        if bc_fn:
            bc_fn(c)
        c.set_fields(*fields)
    a.step(6)
    assert b.step_fused(6)
    for f in ("rho", "U", "p", "e", "rhoE"):
        x, y = a.field(f), b.field(f)
        assert np.abs(x - y).max() <= 1e-14 * np.abs(x).max(), (kind, f, np.abs(x - y).max())
    assert a.info()["steps"] == b.info()["steps"] == 6


def test_fused_path_refuses_what_it_does_not_cover():
    mesh = make_mesh("box654_tri")          # triangles
    c = OracleCase(oracle_mesh_of(mesh), q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    c.set_fields(*cases.box_initial_fields(mesh.array("C").reshape(-1, 3)))
    assert not c.step_fused(1)
    mesh = make_mesh("box654_jitter")       # a qgdFlux wall: its boundary condition is re-evaluated in the middle of the assembly
    c = OracleCase(oracle_mesh_of(mesh), q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    c.set_bc(2, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))
    c.set_fields(*cases.box_initial_fields(mesh.array("C").reshape(-1, 3)))
    assert not c.step_fused(1)
    c = OracleCase(oracle_mesh_of(mesh), q.default_options(stencil="reduced", deltaT=1e-3))
    c.set_fields(*cases.box_initial_fields(mesh.array("C").reshape(-1, 3)))
    assert not c.step_fused(1)
