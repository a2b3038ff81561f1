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


DEV_CASES = [("box654_jitter", "GaussVolPoint"), ("box654_poly", "GaussVolPoint"), ("box654_tri", "reduced"), ("plane2d_jitter", "leastSquares"),
             ("step2d", "GaussVolPoint")]


@pytest.mark.parametrize("mesh_kind,scheme", DEV_CASES)
def test_device_pointer_entries_match_oracle(mesh_kind, scheme):
    """qgd_fvsc_*_dev / qgd_interpolate_dev: fields resident in device memory in, face field in device memory out, same numbers
    as the oracle (and, bit for bit, as the host-pointer entries)"""
    mesh = make_mesh(mesh_kind)
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": scheme}})
    for op, nc in OPS:
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, seed=sum(map(ord, mesh_kind + op)) + 1)
        rc, ref = om.fvsc(scheme, op, cell, bnd)
        assert rc == 0
        nco = 3 * nc if op.startswith("grad") else nc // 3
        vf = q.deviceVolField("f", dev.to_device(cell), dev.to_device(bnd), nc)
        out = dev.alloc(8 * mesh.nFaces * nco)
        ret = fvsc.grad(dev, vf, out=out) if op.startswith("grad") else fvsc.div(dev, vf, out=out)
        assert ret == out
        dev.sync()
        got = dev.to_host(out, ref.shape)
        assert rel_err(got, ref) <= TOL, (mesh_kind, scheme, op, rel_err(got, ref))
        host = fvsc.grad(dev, q.volField("f", cell, bnd)) if op.startswith("grad") else fvsc.div(dev, q.volField("f", cell, bnd))
        assert np.array_equal(got, host), (mesh_kind, scheme, op)
        for ptr in (vf.internal, vf.boundary, out):
            dev.release(ptr)
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 3, 9)
    vf = q.deviceVolField("U", dev.to_device(cell), dev.to_device(bnd), 3)
    out = dev.alloc(8 * mesh.nFaces * 3)
    fvsc.qgdInterpolate(dev, vf, out=out)
    dev.sync()
    assert np.array_equal(dev.to_host(out, (mesh.nFaces, 3)), fvsc.qgdInterpolate(dev, q.volField("U", cell, bnd)))
    dev.close()


def test_device_pointer_qhd_and_species_blocks_match_the_host_entries():
    import ctypes as C
    from qgdsolver_amd import _lib as L
    from qgdsolver_amd import qhdfoam
    from qgdsolver_amd.qgdfoam import speciesFlux
    mesh = make_mesh("box654_poly")
    dev = q.Device(mesh)
    rng = np.random.default_rng(12)
    nc, nb, nf = mesh.nCells, mesh.nBoundaryFaces, mesh.nFaces
    U, T, p, rho = [(rng.standard_normal((nc, k) if k > 1 else nc), rng.standard_normal((nb, k) if k > 1 else nb)) for k in (3, 1, 1, 1)]
    rho = (1.0 + 0.1 * rho[0], 1.0 + 0.1 * rho[1])
    tau, phi = 1e-3 * (1.0 + rng.random(nf)), 1e-3 * rng.standard_normal(nf)
    ref = qhdfoam.updateFluxes(dev, "GaussVolPoint", U, T, rho, tau, 3e-3, (0.0, -9.81, 0.0), p=p, phi=phi)
    ins = qhdfoam.QhdInputs()
    keep = {}
    for name, arr in (("U", U[0]), ("Ub", U[1]), ("T", T[0]), ("Tb", T[1]), ("p", p[0]), ("pb", p[1]), ("rho", rho[0]), ("rhob", rho[1]),
                      ("tauQGDf", tau), ("phi", phi)):
        keep[name] = dev.to_device(arr)
        setattr(ins, name, C.cast(C.c_void_p(keep[name]), L.c_double_p))
    ins.beta = 3e-3
    ins.g[0], ins.g[1], ins.g[2] = 0.0, -9.81, 0.0
    outs = qhdfoam.QhdOutputs()
    widths = dict(gradUf=9, gradTf=3, phiu=1, phiwo=1, taubyrhof=1, gradPf=3, Wf=3, phiUf=3, phiTf=1, phiTauTReg=1)
    ptrs = {}
    for name, w in widths.items():
        ptrs[name] = dev.alloc(8 * nf * w)
        setattr(outs, name, C.cast(C.c_void_p(ptrs[name]), L.c_double_p))
    L.check(L.lib.qgd_qhd_fluxes_dev(dev._h, L.FVSC_GAUSSVOLPOINT, C.byref(ins), C.byref(outs)), "qgd_qhd_fluxes_dev")
    dev.sync()
    for name, w in widths.items():
        got = dev.to_host(ptrs[name], (nf, w) if w > 1 else (nf,))
        assert np.array_equal(got, ref[name]), name
    # species block
    Y = (rng.random(nc), rng.random(nb))
    phiJm = 1e-3 * rng.standard_normal(nf)
    want = speciesFlux(dev, "GaussVolPoint", Y, U, phiJm, phi, tau)
    d = {k: dev.to_device(v) for k, v in dict(Y=Y[0], Yb=Y[1], U=U[0], Ub=U[1], jm=phiJm, phi=phi, tau=tau).items()}
    o = {k: dev.alloc(8 * nf * w) for k, w in dict(phiJmY=1, diffusiveFlux=1, gradYf=3).items()}
    vp = C.c_void_p
    L.check(L.lib.qgd_species_flux_dev(dev._h, L.FVSC_GAUSSVOLPOINT, vp(d["Y"]), vp(d["Yb"]), vp(d["U"]), vp(d["Ub"]), vp(d["jm"]), vp(d["phi"]),
                                       vp(d["tau"]), vp(o["phiJmY"]), vp(o["diffusiveFlux"]), vp(o["gradYf"])), "qgd_species_flux_dev")
    dev.sync()
    for k, w in dict(phiJmY=1, diffusiveFlux=1, gradYf=3).items():
        assert np.array_equal(dev.to_host(o[k], (nf, w) if w > 1 else (nf,)), want[k]), k
    dev.close()
