"""qgdFlux with a divSchemes entry `Gauss upwind` [QGDInterpolate.H L86-104 -> fvc::flux].

QGDFoam: ``div(phiJm,U)`` / ``div(phiJm,H)`` [QGDFoam/updateFluxes.H L78, L119]; QHDFoam: ``div(phi,U)`` / ``div(phi,T)``
[QHDUEqn.H L41, QHDTEqn.H L65].  L0 (OpenFOAM, restated): gaussConvectionScheme::flux = faceFlux * upwind.interpolate(vf),
upwind::weights = pos0(faceFlux), surfaceInterpolationScheme::interpolate = lambda (vf[P] - vf[N]) + vf[N]; patch faces keep
the patch value.

CPU: the oracle's branch is what that says (recomputed here in numpy from the oracle's own fields), is conservative, and is the
linear branch where the two cells agree.  GPU: the face kernels' in-register branch against the oracle on the case matrix of
test_case_parity_gpu.py (<= 1e-10), the implicit branch, QHDFoam in both branches, and the stateless ``qgd_flux_upwind``."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
from qgdsolver_amd import fvsc, qhdfoam

import cases
from oracle import OracleCase, OracleMesh, OracleQhdCase
from util import make_mesh, oracle_mesh_of, rel_err


def upwind_numpy(mesh, flux, cell, bnd):
    own, nei, nif = mesh.array("owner"), mesh.array("neighbour"), mesh.nInternalFaces
    cell = cell.reshape(mesh.nCells, -1)
    bnd = bnd.reshape(mesh.nBoundaryFaces, -1)
    lam = (flux[:nif] >= 0.0).astype(np.float64)[:, None]
    inner = lam * (cell[own[:nif]] - cell[nei]) + cell[nei]
    return np.concatenate([inner, bnd], axis=0)


@pytest.mark.parametrize("kind,scheme", [("box654_jitter", "GaussVolPoint"), ("plane2d_jitter", "leastSquares"), ("box654", "reduced")])
def test_oracle_upwind_branch_is_fvc_flux_with_gauss_upwind(kind, scheme):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    C = mesh.array("C").reshape(-1, 3)
    U, T, p = cases.box_initial_fields(C)
    U[:, 0] += 0.05   # fluxes of both signs
    if mesh.nGeometricD == 2:
        U[:, 2] = 0.0
    types = mesh.array("patchType")

    def make(**kw):
        oc = OracleCase(om, q.default_options(stencil=scheme, deltaT=1e-3, mu=1e-3, **kw))
        for ip in range(mesh.nPatches):
            if types[ip] == L.PATCH_EMPTY:
                oc.set_bc(ip, U=("none", None), T=("none", None), p=("none", None))
        oc.set_fields(U, T, p)
        oc.updateFluxes()
        return oc

    lin, upw = make(), make(fluxSchemeU=1, fluxSchemeH=1)
    phiJm = upw.field("phiJm")
    assert np.array_equal(phiJm, lin.field("phiJm"))               # the mass flux does not depend on the scheme
    assert (phiJm[:mesh.nInternalFaces] > 0).any() and (phiJm[:mesh.nInternalFaces] < 0).any()
    live = np.ones(mesh.nFaces, dtype=bool)
    ps, pz = mesh.array("patchStart"), mesh.array("patchSize")
    for ip in range(mesh.nPatches):
        if types[ip] == L.PATCH_EMPTY:
            live[ps[ip]:ps[ip] + pz[ip]] = False
    Uup = upwind_numpy(mesh, phiJm, upw.field("U"), upw.field("U.boundary"))
    Hup = upwind_numpy(mesh, phiJm, upw.field("H"), upw.field("H.boundary"))[:, 0]
    assert np.array_equal(upw.field("phiJmU")[live], (phiJm[:, None] * Uup)[live])
    assert np.array_equal(upw.field("phiJmH")[live], (phiJm * Hup)[live])
    assert not np.array_equal(upw.field("phiJmU"), lin.field("phiJmU"))
    # everything that is not a qgdFlux is untouched
    for name in ("phiP", "phiPi", "phiQ", "phiPiU", "phiwStar"):
        assert np.array_equal(upw.field(name), lin.field(name)), name
    # one scheme per flux
    only_u = make(fluxSchemeU=1)
    assert np.array_equal(only_u.field("phiJmU"), upw.field("phiJmU")) and np.array_equal(only_u.field("phiJmH"), lin.field("phiJmH"))
    # and the step stays conservative: closed box (zeroGradient everywhere would leak; use the interior balance instead)
    upw.step(5)
    assert np.isfinite(upw.field("rho")).all() and upw.field("rho").min() > 0


def test_oracle_qhd_upwind_branch():
    mesh = make_mesh("box654_jitter")
    om = oracle_mesh_of(mesh)

    def make(**kw):
        opt = qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=1.0, rho0=1.0, mu=1e-2, Pr=0.71, beta=3e-3,
                                  g=(0.0, -9.81, 0.0), deltaT=1e-3, pTol=1e-13, pMaxIter=3000, **kw)
        oc = OracleQhdCase(om, opt)
        for ip in range(mesh.nPatches):
            oc.set_bc(ip, U=("fixedValue", (0.0, 0.0, 0.0)), T=("fixedValue", 310.0 if ip == 0 else 290.0) if ip < 2 else ("zeroGradient", None),
                      p=("zeroGradient", None))
        C = mesh.array("C").reshape(-1, 3)
        rng = np.random.default_rng(5)
        oc.set_fields(0.02 * rng.standard_normal((mesh.nCells, 3)), 300.0 + 10.0 * (0.5 - C[:, 0]), np.zeros(mesh.nCells))
        return oc

    lin, upw = make(), make(fluxSchemeU=1, fluxSchemeT=1)
    lin.step(1); upw.step(1)
    # the pressure equation of the first step does not see the scheme; the U and T equations do
    assert np.array_equal(lin.field("phi"), upw.field("phi"))
    assert not np.array_equal(lin.field("T"), upw.field("T")) and not np.array_equal(lin.field("U"), upw.field("U"))
    # T: the upwind transport is still conservative (walls: no flux through fixedValue-U walls, conduction through the two isothermal ones)
    V = om.array("V")
    assert abs(float((V * (upw.field("T") - lin.field("T"))).sum())) <= 1e-12 * float((V * lin.field("T")).sum())
    upw.step(10)
    assert np.isfinite(upw.field("U")).all()


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
def _case_matrix():
    import test_case_parity_gpu as t
    return t.CASES


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(14))
def test_device_upwind_fluxes_match_the_oracle_on_the_case_matrix(idx):
    import test_case_parity_gpu as t
    mesh_kind, scheme, bc_fn, init_fn, opt = t.CASES[idx]
    opt = dict(opt, fluxSchemeU=1, fluxSchemeH=1)
    mesh, dev, gc, oc = t.build_pair(mesh_kind, scheme, bc_fn, init_fn, **opt)
    gc.updateFluxes(); oc.updateFluxes()
    t.compare_fields(gc, oc, t.FACE_FIELDS, t.FLUX_TOL, (mesh_kind, scheme, "upwind fluxes"))
    # the branch is live: it differs from the linear product on some internal face
    nif = mesh.nInternalFaces
    Uf = fvsc.qgdInterpolate(dev, fvsc.volField("U", gc.field("U"), gc.field("U.boundary")))
    Hf = fvsc.qgdInterpolate(dev, fvsc.volField("H", gc.field("H"), gc.field("H.boundary")))
    assert (not np.array_equal(gc.field("phiJmU")[:nif], (gc.field("phiJm")[:, None] * Uf)[:nif])      # (a uniform U upwinds to itself: step2d)
            or not np.array_equal(gc.field("phiJmH")[:nif], (gc.field("phiJm") * Hf)[:nif]))
    for chunk in (1, 9):
        gc.step(chunk); oc.step(chunk)
        t.compare_fields(gc, oc, ["rho", "U", "p", "e", "rhoU", "rhoE"], t.STATE_TOL, (mesh_kind, scheme, f"upwind step+{chunk}"))
    gc.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("u,h", [(1, 0), (0, 1)])
def test_device_one_upwind_flux_at_a_time(u, h):
    import test_case_parity_gpu as t
    mesh, dev, gc, oc = t.build_pair("box654_jitter", "GaussVolPoint", t.mixed_box_bcs, None, deltaT=5e-4, mu=2e-3, fluxSchemeU=u, fluxSchemeH=h)
    gc.step(8); oc.step(8)
    t.compare_fields(gc, oc, ["rho", "U", "p", "e"], t.STATE_TOL, ("one flux", u, h))
    gc.close(); dev.close()


@pytest.mark.gpu
def test_device_upwind_on_the_face_tiles_equals_the_gather_kernel():
    """a box large enough for the LDS-staged tile kernel (its upwind instantiation) against the oracle"""
    mesh = q.PolyMesh.box(24, 20, 16)
    om = OracleMesh(mesh.primitives())
    opt = q.default_options(stencil="GaussVolPoint", deltaT=5e-4, mu=1e-3, fluxSchemeU=1, fluxSchemeH=1)
    dev = q.Device(mesh)
    assert dev.face_tiles()["facesPerTile"] > 0
    gc, oc = q.QGDFoamCase(dev, opt), OracleCase(om, opt)
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    gc.set_fields(U, T, p); oc.set_fields(U, T, p)
    gc.step(6); oc.step(6)
    for name in ("rho", "U", "p", "e"):
        assert rel_err(gc.field(name), oc.field(name)) <= 1e-10, name
    gc.close(); dev.close()


@pytest.mark.gpu
def test_device_upwind_with_implicit_diffusion():
    import test_case_parity_gpu as t
    mesh, dev, gc, oc = t.build_pair("box654_jitter", "GaussVolPoint", t.mixed_box_bcs, None, deltaT=5e-4, mu=2e-3, implicitDiffusion=1,
                                     implicitTol=1e-14, fluxSchemeU=1, fluxSchemeH=1)
    gc.step(8); oc.step(8)
    t.compare_fields(gc, oc, ["rho", "U", "p", "e"], t.STATE_TOL, ("implicit upwind",))
    gc.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,stencil,implicit", [("box654_jitter", "GaussVolPoint", 0), ("box654_jitter", "GaussVolPoint", 1),
                                                   ("plane2d_jitter", "leastSquares", 0), ("box654", "reduced", 1)])
def test_device_qhd_upwind_matches_the_oracle(kind, stencil, implicit):
    import test_qhd_case as t
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    opt = t.options(stencil, deltaT=1e-3, implicitDiffusion=implicit, implicitTol=1e-13, fluxSchemeU=1, fluxSchemeT=1)
    dev = q.Device(mesh)
    gc, oc = qhdfoam.QHDFoamCase(dev, opt), OracleQhdCase(om, opt)
    U, T, p = t.initial(mesh)
    rng = np.random.default_rng(3)
    U = U + 1e-2 * rng.standard_normal(U.shape)
    if mesh.nGeometricD == 2:
        U[:, 2] = 0.0
    for c in (gc, oc):
        t.cavity_bcs(c, mesh)
        c.set_fields(U, T, p)
    gc.step(12); oc.step(12)
    for f in ("U", "T", "p", "phi"):
        ref = oc.field(f)
        assert np.abs(gc.field(f) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300), (kind, stencil, implicit, f)
    gc.close(); dev.close()


@pytest.mark.gpu
def test_stateless_flux_upwind_entry():
    mesh = make_mesh("box654_poly")
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}, "divSchemes": {"div(phiJm,U)": ["Gauss", "upwind"], "div(phiJm,H)": "Gauss linear"}})
    rng = np.random.default_rng(1)
    flux = rng.standard_normal(mesh.nFaces)
    Uc, Ub = rng.standard_normal((mesh.nCells, 3)), rng.standard_normal((mesh.nBoundaryFaces, 3))
    U = fvsc.volField("U", Uc, Ub)
    Uf = fvsc.qgdInterpolate(dev, U)
    got = fvsc.qgdFlux(dev, flux, Uf, psi=U, flux_name="div(phiJm,U)")
    assert np.array_equal(got, flux[:, None] * upwind_numpy(mesh, flux, Uc, Ub))
    H = fvsc.volField("H", Uc[:, 0].copy(), Ub[:, 0].copy())
    Hf = fvsc.qgdInterpolate(dev, H)
    assert np.array_equal(fvsc.qgdFlux(dev, flux, Hf, psi=H, flux_name="div(phiJm,H)"), flux * Hf)     # Gauss linear = flux*psif
    dev.fvSchemes["divSchemes"]["div(phiJm,H)"] = ["Gauss", "vanLeer"]
    with pytest.raises(L.QgdError):
        fvsc.qgdFlux(dev, flux, Hf, psi=H, flux_name="div(phiJm,H)")
    dev.fvSchemes["interpolationSchemes"] = {"default": "linear", "interpolate(H)": "linear"}
    assert np.array_equal(fvsc.qgdInterpolate(dev, H), Hf)
    dev.fvSchemes["interpolationSchemes"] = {"default": "cubic"}
    with pytest.raises(L.QgdError):
        fvsc.qgdInterpolate(dev, H)
    dev.close()
