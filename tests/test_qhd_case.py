"""QHDFoam case (explicit branch of QHDFoam.C L83-139): properties of the oracle on CPU, device parity on the GPU.

CPU (oracle): a fluid at rest with uniform T and g = 0 stays at rest; the face flux phi = phiu - phiwo + pEqn.flux() is
divergence free after every pressure solve; with impermeable walls (qhdFlux fed by the registered flux) the wall fluxes
vanish; T is transported conservatively (sum V T changes only through wall conduction); a buoyant cavity starts to turn the
way round the listing's body force BdFrc = +beta*T*g implies.  GPU: the device case against the oracle after N steps on small meshes of every stencil kind, the
multigrid-preconditioned pressure solver against the Jacobi one, and the 2 M / 8 M-cell pressure equation in < 100
iterations."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam

from oracle import OracleQhdCase
from util import make_mesh, oracle_mesh_of

WALL = dict(U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("qhdFluxCoupled", None))


def cavity_bcs(case, mesh, hot=310.0, cold=290.0):
    """patches 0/1 (xMin/xMax) hot/cold walls, the others adiabatic walls; empty patches stay empty"""
    types = mesh.array("patchType")
    for ip in range(mesh.nPatches):
        if types[ip] == q._lib.PATCH_EMPTY:
            case.set_bc(ip, U=("none", None), T=("none", None), p=("none", None))
        elif ip == 0:
            case.set_bc(ip, U=WALL["U"], T=("fixedValue", hot), p=WALL["p"])
        elif ip == 1:
            case.set_bc(ip, U=WALL["U"], T=("fixedValue", cold), p=WALL["p"])
        else:
            case.set_bc(ip, **WALL)


def options(stencil="GaussVolPoint", **kw):
    base = dict(stencil=stencil, tauModel="HbyUQHD", aQGD=0.5, UQHD=1.0, rho0=1.0, mu=1e-2, Pr=0.71, beta=3e-3, g=(0.0, -9.81, 0.0),
                deltaT=2e-3, pTol=1e-12, pMaxIter=3000, pRefCell=0, pRefValue=0.0, precond=1)
    base.update(kw)
    return qhdfoam.qhd_options(**base)


def initial(mesh):
    C = mesh.array("C").reshape(-1, 3)
    n = mesh.nCells
    return np.zeros((n, 3)), 300.0 + 10.0 * (0.5 - C[:, 0]), np.zeros(n)


def divergence(mesh, phi):
    own, nei, nif = mesh.array("owner"), mesh.array("neighbour"), mesh.nInternalFaces
    div = np.zeros(mesh.nCells)
    np.add.at(div, own, phi)
    np.subtract.at(div, nei, phi[:nif])
    return div


def test_oracle_rest_state_and_divergence_free_flux():
    mesh = make_mesh("box654_jitter")
    om = oracle_mesh_of(mesh)
    oc = OracleQhdCase(om, options(g=(0.0, 0.0, 0.0)))
    cavity_bcs(oc, mesh, hot=300.0, cold=300.0)
    n = mesh.nCells
    oc.set_fields(np.zeros((n, 3)), np.full(n, 300.0), np.zeros(n))
    oc.step(5)
    assert np.abs(oc.field("U")).max() == 0.0 and np.all(oc.field("T") == 300.0) and np.abs(oc.field("p")).max() <= 1e-300
    # buoyant cavity
    oc = OracleQhdCase(om, options())
    cavity_bcs(oc, mesh)
    U, T, p = initial(mesh)
    oc.set_fields(U, T, p)
    V = om.array("V")
    heat0 = float((V * oc.field("T")).sum())
    oc.step(20)
    phi = oc.field("phi")
    assert np.abs(divergence(mesh, phi)).max() <= 1e-9 * np.abs(phi).max()
    assert np.abs(phi[mesh.nInternalFaces:]).max() <= 1e-12 * np.abs(phi).max()     # impermeable walls
    # the body force is BdFrc = +beta*T*g as listed [QHDFoam/updateFields.H L66]: warmer fluid is pushed ALONG g, so with g
    # along -y the fluid at the hot wall (xMin) moves down and rises at the cold wall (a case file gives g with that in mind)
    C = mesh.array("C").reshape(-1, 3)
    Uy = oc.field("U")[:, 1]
    assert Uy[C[:, 0] < 0.2].mean() < 0 < Uy[C[:, 0] > 0.8].mean()
    # T: conservative transport + conduction through the two isothermal walls only
    heat1 = float((V * oc.field("T")).sum())
    assert abs(heat1 - heat0) < 1e-3 * abs(heat0) and heat1 != heat0
    assert oc.info()["pIterations"] > 0 and oc.info()["pFinalResidual"] < 1e-12


GPU_CASES = [("box654_jitter", "GaussVolPoint"), ("box654_tri", "GaussVolPoint"), ("box654_poly", "GaussVolPoint"), ("box654", "reduced"),
             ("plane2d_jitter", "leastSquares"), ("plane2d", "GaussVolPoint"), ("step2d", "leastSquares")]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,stencil", GPU_CASES)
def test_device_qhd_case_matches_oracle(kind, stencil):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    opt = options(stencil, deltaT=1e-3)
    dev = q.Device(mesh)
    gc, oc = qhdfoam.QHDFoamCase(dev, opt), OracleQhdCase(om, opt)
    U, T, p = initial(mesh)
    rng = np.random.default_rng(3)
    U = U + 1e-2 * rng.standard_normal(U.shape)
    if mesh.nGeometricD == 2:
        U[:, 2] = 0.0
    for c in (gc, oc):
        cavity_bcs(c, mesh)
        c.set_fields(U, T, p)
    gc.step(15); oc.step(15)
    for f in ("U", "T", "p", "phi", "U.boundary", "T.boundary", "p.boundary"):
        ref = oc.field(f)
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(gc.field(f) - ref).max() <= 1e-9 * scale, (kind, stencil, f, np.abs(gc.field(f) - ref).max() / scale)
    assert np.abs(oc.field("U")).max() > 1e-3
    gi = gc.info()
    assert gi["pFinalResidual"] < 1e-12 and gi["steps"] == 15
    gc.close(); dev.close()


@pytest.mark.gpu
def test_fixed_value_pressure_and_fixed_gradient_patches():
    """p fixedValue on one patch (no reference level), fixedGradient (what qhdFlux is inside QHDFoam) on another, slip walls"""
    mesh = make_mesh("box654_jitter")
    om = oracle_mesh_of(mesh)
    opt = options("GaussVolPoint", tauModel="T0byGr", T0=1.0, Gr=200.0, deltaT=1e-3)
    dev = q.Device(mesh)
    gc, oc = qhdfoam.QHDFoamCase(dev, opt), OracleQhdCase(om, opt)
    U, T, p = initial(mesh)
    U[:, 0] = 0.05
    for c in (gc, oc):
        c.set_bc(0, U=("fixedValue", (0.05, 0.0, 0.0)), T=("fixedValue", 305.0), p=("zeroGradient", None))
        c.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 0.0))
        c.set_bc(2, U=("slip", None), T=("zeroGradient", None), p=("fixedGradient", 0.02))
        for ip in (3, 4, 5):
            c.set_bc(ip, U=("slip", None), T=("zeroGradient", None), p=("zeroGradient", None))
        c.set_fields(U, T, p)
    gc.step(12); oc.step(12)
    for f in ("U", "T", "p", "phi"):
        ref = oc.field(f)
        assert np.abs(gc.field(f) - ref).max() <= 1e-9 * np.abs(ref).max(), f
    gc.close(); dev.close()


@pytest.mark.gpu
def test_multigrid_and_jacobi_preconditioners_agree():
    mesh = q.PolyMesh.box(20, 18, 16).jitter(0.15, seed=5)
    dev = q.Device(mesh)
    res = {}
    for precond in (0, 1):
        c = qhdfoam.QHDFoamCase(dev, options(precond=precond, deltaT=1e-3))
        cavity_bcs(c, mesh)
        c.set_fields(*initial(mesh))
        c.step(5)
        res[precond] = (c.field("p"), c.field("U"), c.info())
        c.close()
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-9 * np.abs(res[0][0]).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-9 * np.abs(res[0][1]).max()
    # 5760 cells: the smoothed-aggregation hierarchy is 5760 -> ~700 rows (solved exactly); QGD_MG_SA=0 builds >= 3 plain levels
    assert res[1][2]["mgLevels"] >= 2, res[1][2]
    assert res[1][2]["pIterations"] < res[0][2]["pIterations"] / 3, (res[0][2], res[1][2])
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("implicit", [0, 1])
def test_staged_face_passes_are_the_generic_walk_bit_for_bit(implicit, monkeypatch):
    """qhdFace1/2TileKernel (the internal faces of a 128-face tile out of LDS, QGD_QHD_TILES default 1) against the generic
    qhdFace1/2Kernel (QGD_QHD_TILES=0): same expressions on the same values -- U, T, p and phi after 5 steps agree bit for bit on
    hexahedra (with a ragged last tile and row ends beyond the caps), on a jittered mesh with triangles and polygon faces, and on a
    scrambled numbering (most tiles left to the generic kernel)."""
    from test_config5_gpu import c5_mesh
    scr = q.PolyMesh.box(12, 10, 8)
    scr.renumber(np.random.default_rng(5).permutation(scr.nCells).astype(np.int32))
    for tag, mesh in (("hex 37x11x5", q.PolyMesh.box(37, 11, 5)), ("hex 20^3 jittered", q.PolyMesh.box(20, 20, 20).jitter(0.15, seed=4)),
                      ("triangles + polygons, Morton order", c5_mesh(16, 8 ** 3, poly=True)), ("scrambled labels", scr)):
        res = {}
        for tiles in (0, 1):
            monkeypatch.setenv("QGD_QHD_TILES", str(tiles))
            dev = q.Device(mesh)
            ft = dev.face_tiles()
            c = qhdfoam.QHDFoamCase(dev, options(deltaT=1e-3, pTol=1e-10, implicitDiffusion=implicit))
            cavity_bcs(c, mesh)
            c.set_fields(*initial(mesh))
            c.step(5)
            res[tiles] = {k: c.field(k) for k in ("U", "T", "p", "phi")}
            c.close(); dev.close()
        if not tag.startswith("scrambled"):
            assert ft["facesPerTile"] == 128 and ft["gatherTiles"] < ft["tiles"], (tag, ft)
        for k in res[0]:
            assert np.isfinite(res[0][k]).all() and np.abs(res[0][k]).max() > 0
            assert np.array_equal(res[0][k], res[1][k]), (tag, k, np.abs(res[0][k] - res[1][k]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("nu0", [0, 1])
def test_fused_cycle_hand_over_changes_no_bit_of_a_solve(nu0, monkeypatch):
    """QGD_MG_FUSE (default 1): the head of the single-precision cycle rides in the CG's axpy kernel, its last post-smoothing sweep
    writes z in double and the block partials of r.z.  Same arithmetic, same partial sums: pressure, velocity and the iteration
    count after 6 steps are those of the separate passes, bit for bit (48^3 cells: three levels; also with one sweep on level 0)."""
    mesh = q.PolyMesh.box(48, 48, 48).jitter(0.1, seed=2)
    dev = q.Device(mesh)
    res = {}
    if nu0:
        monkeypatch.setenv("QGD_MG_NU0", str(nu0))
    for fuse in (0, 1):
        monkeypatch.setenv("QGD_MG_FUSE", str(fuse))
        c = qhdfoam.QHDFoamCase(dev, options(deltaT=1e-3, pTol=1e-10))
        cavity_bcs(c, mesh)
        c.set_fields(*initial(mesh))
        its = []
        for _ in range(6):
            c.step(1)
            its.append(c.info()["pIterations"])
        res[fuse] = (c.field("p"), c.field("U"), c.field("T"), its, c.info())
        c.close()
    assert res[1][4]["mgLevels"] >= 3, res[1][4]
    assert res[0][3] == res[1][3] and min(res[0][3]) >= 2, (res[0][3], res[1][3])
    for k in range(3):
        assert np.isfinite(res[0][k]).all() and np.array_equal(res[0][k], res[1][k]), k
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 200])
def test_pressure_equation_at_scale_under_100_iterations(n):
    """2 M and 8 M cells (one 8-GPU shard of config 5's size): QHDpEqn.H L36-47 to 1e-8 in < 100 iterations"""
    mesh = q.PolyMesh.box(n, n, n)
    dev = q.Device(mesh)
    c = qhdfoam.QHDFoamCase(dev, options(deltaT=0.2 / n, pTol=1e-8, pMaxIter=400))
    cavity_bcs(c, mesh)
    c.set_fields(*initial(mesh))
    c.step(2)
    info = c.info()
    assert info["pFinalResidual"] < 1e-8 and 0 < info["pIterations"] < 100, info
    phi = c.field("phi")
    div = divergence(mesh, phi)
    assert np.abs(div).max() <= 1e-5 * np.abs(phi).max()
    assert np.isfinite(c.field("U")).all() and np.abs(c.field("T") - 300).max() <= 10.0 + 1e-9
    print(f"QHD {n}^3: {info}")
    c.close(); dev.close()
