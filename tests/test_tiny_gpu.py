"""Degenerate sizes: one cell (no internal face at all), two cells (one internal face), a single row -- through the whole
path (fvsc operators, flux assembly, steps) against the oracle."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
from oracle import OracleCase
from util import oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu
G, E = L.PATCH_GENERIC, L.PATCH_EMPTY


@pytest.mark.parametrize("implicit", [0, 1])
@pytest.mark.parametrize("dims,ptypes,stencil", [
    ((1, 1, 1), None, "GaussVolPoint"),
    ((1, 1, 1), None, "reduced"),
    ((2, 1, 1), None, "GaussVolPoint"),
    ((3, 1, 1), [G, G, E, E, E, E], "GaussVolPoint"),
    ((2, 2, 1), [G, G, G, G, E, E], "leastSquares"),
    ((2, 2, 1), [G, G, G, G, E, E], "GaussVolPoint"),
])
def test_tiny_meshes(dims, ptypes, stencil, implicit):
    mesh = q.PolyMesh.box(*dims, patch_types=ptypes)
    assert mesh.nInternalFaces == (dims[0] - 1) * dims[1] * dims[2] + dims[0] * (dims[1] - 1) * dims[2] + dims[0] * dims[1] * (dims[2] - 1)
    n = mesh.nCells
    rng = np.random.default_rng(1)
    U = 0.1 * rng.standard_normal((n, 3))
    if ptypes:
        for d in range(3):
            if ptypes[2 * d] == E:
                U[:, d] = 0.0
    T = 1.0 + 0.05 * rng.standard_normal(n)
    p = 1.0 + 0.05 * rng.standard_normal(n)
    opt = q.default_options(stencil=stencil, deltaT=1e-3, mu=1e-3, implicitDiffusion=implicit, implicitTol=1e-14, implicitMaxIter=200)
    dev = q.Device(mesh)
    gc = q.QGDFoamCase(dev, opt)
    oc = OracleCase(oracle_mesh_of(mesh), opt)
    for c in (gc, oc):
        c.set_bc(0, U=("fixedValue", (0.1, 0.0, 0.0)), T=("fixedValue", 1.02), p=("zeroGradient", None))
        c.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
        c.set_fields(U, T, p)
    gc.updateFluxes(); oc.updateFluxes()
    for f in ("phiJm", "phiJmU", "phiPi", "phiQ", "gradUf", "gradPf"):
        assert rel_err(gc.field(f), oc.field(f)) <= 1e-11, (dims, stencil, f)
    gc.step(10); oc.step(10)
    for f in ("rho", "U", "p", "e"):
        assert rel_err(gc.field(f), oc.field(f)) <= 1e-11, (dims, stencil, implicit, f)
    assert np.all(np.isfinite(gc.field("rho")))
    gc.close(); dev.close()


@pytest.mark.parametrize("dims,ptypes", [((1, 1, 1), None), ((2, 1, 1), None), ((3, 2, 1), [G, G, G, G, E, E]), ((2, 2, 2), None)])
def test_tiny_meshes_qhd(dims, ptypes):
    """QHDFoam's step where the pressure equation has one to eight rows (multigrid has nothing to coarsen: Jacobi-PCG or one level)"""
    from qgdsolver_amd import qhdfoam
    from oracle import OracleQhdCase
    mesh = q.PolyMesh.box(*dims, patch_types=ptypes)
    n = mesh.nCells
    rng = np.random.default_rng(2)
    U = 1e-2 * rng.standard_normal((n, 3))
    if ptypes:
        U[:, 2] = 0.0
    T = 300.0 + rng.standard_normal(n)
    opt = qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=1.0, rho0=1.0, mu=1e-2, Pr=0.71, beta=3e-3,
                              g=(0.0, -9.81, 0.0), deltaT=1e-3, pTol=1e-13, pMaxIter=500, pRefCell=0, pRefValue=0.0, precond=1)
    dev = q.Device(mesh)
    gc, oc = qhdfoam.QHDFoamCase(dev, opt), OracleQhdCase(oracle_mesh_of(mesh), opt)
    types = mesh.array("patchType")
    for c in (gc, oc):
        for ip in range(mesh.nPatches):
            if types[ip] == E:
                c.set_bc(ip, U=("none", None), T=("none", None), p=("none", None))
            else:
                c.set_bc(ip, U=("fixedValue", (0.0, 0.0, 0.0)), T=("fixedValue", 301.0) if ip == 0 else ("zeroGradient", None), p=("qhdFluxCoupled", None))
        c.set_fields(U, T, np.zeros(n))
    gc.step(8); oc.step(8)
    for f in ("U", "T", "p", "phi"):
        ref = oc.field(f)
        # (+ 1e-15: on a closed one-row mesh phi IS rounding noise, 1e-17 of a flux scale of 1e-3, and that noise depends on where the
        # pressure solve starts)
        assert np.abs(gc.field(f) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30) + 1e-15, (dims, f)
    gc.close(); dev.close()


def test_a_refused_hip_call_does_not_poison_the_next_one():
    """An entry that fails inside HIP reports that error once; the launches of later calls must not find it in hipGetLastError()
    (seen: `invalid device ordinal` of a refused qgd_device_create surfacing in the next qgd_case_create)."""
    mesh = q.PolyMesh.box(4, 3, 2)
    with pytest.raises(q.QgdError):
        q.Device(mesh, device_id=4096)
    dev = q.Device(mesh)
    n = mesh.nCells
    for impl in (1, 0):
        case = q.QGDFoamCase(dev, q.default_options(deltaT=1e-3, mu=1e-3, implicitDiffusion=impl))
        case.set_fields(np.zeros((n, 3)), np.ones(n), np.ones(n))
        case.step(2)
        assert np.isfinite(case.field("rho")).all()
        case.close()
    from qgdsolver_amd import qhdfoam
    qc = qhdfoam.QHDFoamCase(dev, qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=1.0, rho0=1.0, mu=1e-2, Pr=0.71,
                                                     beta=3e-3, g=(0.0, -9.81, 0.0), deltaT=1e-3, pTol=1e-10, pMaxIter=200, pRefCell=0, pRefValue=0.0))
    qc.close(); dev.close()


def test_cases_and_their_device_may_die_in_any_order():
    """A case points at its device inside the library.  Under reference counting the case goes first; the cycle collector
    finalises in any order (seen: qgd_case_free reading a freed device and leaving `invalid device ordinal` behind for the next
    launch check).  The device frees the cases still open on it before itself; both close() are idempotent."""
    import ctypes
    import gc
    from qgdsolver_amd import qhdfoam
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGetLastError()
    mesh = q.PolyMesh.box(4, 3, 2)
    opt = q.default_options(deltaT=1e-3, mu=1e-3, implicitDiffusion=1)
    qopt = qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=1.0, rho0=1.0, mu=1e-2, Pr=0.71, beta=3e-3,
                               g=(0.0, -9.81, 0.0), deltaT=1e-3, pTol=1e-10, pMaxIter=200, pRefCell=0, pRefValue=0.0)
    dev = q.Device(mesh)
    case, qcase = q.QGDFoamCase(dev, opt), qhdfoam.QHDFoamCase(dev, qopt)
    dev.close()                                    # the device first
    assert case._h is None and qcase._h is None    # ... has freed its cases
    case.close(); qcase.close()
    assert hip.hipGetLastError() == 0
    for _ in range(3):                             # one garbage cycle holding both: the collector picks the order
        dev = q.Device(mesh)
        cyc = [dev, q.QGDFoamCase(dev, opt), qhdfoam.QHDFoamCase(dev, qopt)]
        cyc.append(cyc)
        del dev, cyc
        gc.collect()
        assert hip.hipGetLastError() == 0
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, opt)
    n = mesh.nCells
    case.set_fields(np.zeros((n, 3)), np.ones(n), np.ones(n))
    case.step(2)
    assert np.isfinite(case.field("rho")).all()
    case.close(); dev.close()


def test_device_free_is_refused_while_its_cases_are_open():
    """the C-ABI's own guard (a C host has no Python mirror ordering its frees): qgd_device_free returns QGD_ERR_INVALID and frees
    nothing while a case created on the device is open"""
    mesh = q.PolyMesh.box(3, 2, 2)
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(deltaT=1e-3))
    rc = L.lib.qgd_device_free(dev._h)
    assert rc != L.QGD_OK and b"still open" in L.lib.qgd_last_error()
    n = mesh.nCells
    case.set_fields(np.zeros((n, 3)), np.ones(n), np.ones(n))
    case.step(1)                      # device and case are untouched
    assert np.isfinite(case.field("rho")).all()
    case.close()
    assert L.lib.qgd_device_free(dev._h) == L.QGD_OK
    dev._h = None


def test_device_close_keeps_its_handle_when_the_free_is_refused():
    """Device.close() with a case it does not know about (created through the raw C entry, never adopted): the library refuses the free,
    close() raises with the library's message and keeps the handle, so the free can be retried once the case is gone."""
    import ctypes as C
    mesh = q.PolyMesh.box(3, 2, 2)
    dev = q.Device(mesh)
    opt = q.default_options(deltaT=1e-3)
    raw = C.c_void_p()
    assert L.lib.qgd_case_create(dev._h, C.byref(opt), C.byref(raw)) == L.QGD_OK
    with pytest.raises(RuntimeError, match="still open"):
        dev.close()
    assert dev._h                     # not leaked: still there to be freed
    assert L.lib.qgd_case_free(raw) == L.QGD_OK
    dev.close()
    assert dev._h is None
