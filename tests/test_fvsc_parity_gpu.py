"""GPU parity of the four fvsc operators (through the C-ABI) against the CPU oracle.

Tolerance: 1e-12 relative to the output field's max magnitude (fp64; the only arithmetic
differences are FMA contraction and the reciprocal of the Gauss volume).
"""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import fvsc

import cases
from util import make_mesh, oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-12
OPS = [("grad_s", 1), ("grad_v", 3), ("div_v", 3), ("div_t", 9)]

MESH_SCHEMES = [
    ("box654", "reduced"), ("box654", "GaussVolPoint"),
    ("box654_jitter", "reduced"), ("box654_jitter", "GaussVolPoint"),
    ("box654_tri", "GaussVolPoint"), ("box654_tri", "reduced"),
    ("box654_poly", "GaussVolPoint"), ("box654_poly", "reduced"),
    ("plane2d", "GaussVolPoint"), ("plane2d", "leastSquares"), ("plane2d", "leastSquaresOpt"), ("plane2d", "reduced"),
    ("plane2d_jitter", "GaussVolPoint"), ("plane2d_jitter", "leastSquares"),
    ("plane2d_y", "GaussVolPoint"), ("plane2d_y", "leastSquares"),
    ("line1d", "GaussVolPoint"), ("line1d", "leastSquares"), ("line1d", "reduced"),
    ("step2d", "GaussVolPoint"), ("step2d", "leastSquares"),
    ("box_sym", "leastSquares"), ("box_sym", "GaussVolPoint"),
]


@pytest.mark.parametrize("mesh_kind,scheme", MESH_SCHEMES)
def test_fvsc_operators_match_oracle(mesh_kind, scheme):
    mesh = make_mesh(mesh_kind)
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": scheme}})
    for op, nc in OPS:
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, seed=sum(map(ord, mesh_kind + op)))
        rc, ref = om.fvsc(scheme, op, cell, bnd)
        assert rc == 0
        vf = q.volField("f", cell, bnd)
        got = fvsc.grad(dev, vf) if op.startswith("grad") else fvsc.div(dev, vf)
        assert got.shape == ref.shape
        assert rel_err(got, ref) <= TOL, (mesh_kind, scheme, op, rel_err(got, ref))
    dev.close()


def test_leastsquares_refused_in_3d():
    """fvsc.C L60-63: leastSquares / leastSquaresOpt are fatal when nGeometricD == 3."""
    mesh = make_mesh("box654")
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "leastSquares"}})
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 1)
    with pytest.raises(q.QgdError) as ei:
        fvsc.grad(dev, q.volField("p", cell, bnd))
    assert ei.value.code == q._lib.ERR_SCHEME
    rc, _ = oracle_mesh_of(mesh).fvsc("leastSquares", "grad_s", cell, bnd)
    assert rc == -4
    dev.close()


def test_unknown_scheme_word():
    mesh = make_mesh("box654")
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "noSuchStencil"}})
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 1)
    with pytest.raises(q.QgdError) as ei:
        fvsc.grad(dev, q.volField("p", cell, bnd))
    assert ei.value.code == q._lib.ERR_UNKNOWN_NAME
    dev.close()


def test_per_term_scheme_entry():
    """fvsc.C L51-58: a grad(<name>) entry overrides default."""
    mesh = make_mesh("box654")
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint", "grad(T)": "reduced"}})
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 5)
    got_T = fvsc.grad(dev, q.volField("T", cell, bnd))
    got_p = fvsc.grad(dev, q.volField("p", cell, bnd))
    assert rel_err(got_T, om.fvsc("reduced", "grad_s", cell, bnd)[1]) <= TOL
    assert rel_err(got_p, om.fvsc("GaussVolPoint", "grad_s", cell, bnd)[1]) <= TOL
    assert set(dev._registry) == {"reduced", "GaussVolPoint"}  # lookupOrNew caches one stencil per word
    dev.close()


def test_gaussvolpoint_refused_on_wedge_meshes_with_prisms():
    """fvsc.C L65-82: fatal for GaussVolPoint when the mesh has wedge patches and prism cells"""
    from test_oracle_properties import wedge_prism_mesh
    mesh = wedge_prism_mesh(True)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint", "grad(r)": "reduced"}})
    with pytest.raises(q.QgdError) as ei:
        fvsc.grad(dev, q.volField("p", np.ones(1), np.ones(5)))
    assert ei.value.code == q._lib.ERR_SCHEME and "wedge" in str(ei.value)
    with pytest.raises(q.QgdError):
        q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint"))
    fvsc.grad(dev, q.volField("r", np.ones(1), np.ones(5)))   # the other stencils are accepted
    dev.close()
    dev2 = q.Device(wedge_prism_mesh(False))
    fvsc.grad(dev2, q.volField("p", np.ones(1), np.ones(5)))   # no wedge patch: accepted
    dev2.close()
