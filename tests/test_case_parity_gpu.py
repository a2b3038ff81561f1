"""GPU parity of the QGDFoam flux assembly and explicit step against the CPU oracle.

Bar (north_star): <= 1e-10 relative on rho, U, p after N steps.  Face fluxes straight after
updateFluxes are held to 1e-11 relative to each field's max magnitude.
"""
import numpy as np
import pytest

import qgdsolver_amd as q

import cases
from oracle import OracleCase
from util import assert_path, device_pair_arms, expects_fused, make_mesh, oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu

STATE_TOL = 1e-10
FLUX_TOL = 1e-11
FACE_FIELDS = ["phiJm", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU", "phiwStar", "phi", "tauQGDf", "hQGDf",
               "gradUf", "gradef", "gradRhof", "gradPf"]
CELL_FIELDS = ["rho", "U", "p", "e", "T", "rhoU", "rhoE", "c", "psi", "mu", "alphau", "tauQGD", "muQGD", "alphauQGD", "hQGD", "H"]


def build_pair(mesh_kind, scheme, bc_fn=None, init_fn=None, arm=None, **opt):
    """arm: "fused" = the one-launch step (fusedFaceCellKernel: the kernel bench.py times), "kernels" = the separate vertex / face / cell
    kernels; None = whichever the library's default picks.  The arm asked for is asserted to be the path that runs."""
    mesh = make_mesh(mesh_kind)
    om = oracle_mesh_of(mesh)
    options = q.default_options(stencil=scheme, **opt)
    dev = q.Device(mesh, fused_tables={None: True, "fused": "any", "fusedAdjust": "any", "kernels": False}[arm])
    gc = q.QGDFoamCase(dev, options)
    if arm is not None:
        assert_path(gc, arm, (mesh_kind, scheme))
    oc = OracleCase(om, options)
    if bc_fn:
        bc_fn(gc)
        bc_fn(oc)
    C = mesh.array("C").reshape(-1, 3)
    U, T, p = (init_fn or cases.box_initial_fields)(C)
    gc.set_fields(U, T, p)
    oc.set_fields(U, T, p)
    return mesh, dev, gc, oc


def compare_fields(gc, oc, names, tol, tag):
    worst = {}
    for n in names:
        e = rel_err(gc.field(n), oc.field(n))
        worst[n] = e
        assert e <= tol, (tag, n, e)
    return worst


def step_init(C):
    n = C.shape[0]
    U = np.zeros((n, 3))
    U[:, 0] = 3.0
    # a smooth perturbation so that every term of the flux algebra is exercised
    T = 1.0 + 0.05 * np.sin(2.0 * C[:, 0]) * np.cos(3.0 * C[:, 1])
    p = 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])
    return U, T, p


def plane_init(C):
    """box_initial_fields with the pressure pulse centred in the (one cell thick) plane"""
    U, T, _ = cases.box_initial_fields(C)
    r2 = (C[:, 0] - 0.5) ** 2 + (C[:, 1] - 0.4) ** 2
    return U, T, 1.0 + 0.1 * np.exp(-r2 / 0.01)


def empty_z_bcs(case):
    for patch in (4, 5):
        case.set_bc(patch, U=("none", None), T=("none", None), p=("none", None))


def mixed_box_bcs(case):
    """inlet fixedValue / outlet zeroGradient / slip walls with qgdFlux / fixed-T wall"""
    case.set_bc(0, U=("fixedValue", (0.3, 0.0, 0.0)), T=("fixedValue", 1.0), p=("zeroGradient", None))
    case.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
    case.set_bc(2, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))
    case.set_bc(3, U=("slip", None), T=("fixedValue", 1.05), p=("qgdFlux", None))
    case.set_bc(4, U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("zeroGradient", None))
    case.set_bc(5, U=("slip", None), T=("zeroGradient", None), p=("zeroGradient", None))


CASES = [
    ("box654", "GaussVolPoint", None, None, dict(deltaT=2e-3, mu=1e-3)),
    ("box654", "reduced", None, None, dict(deltaT=2e-3, mu=1e-3)),
    ("box654_jitter", "GaussVolPoint", None, None, dict(deltaT=1e-3, mu=1e-3)),
    ("box654_tri", "GaussVolPoint", None, None, dict(deltaT=1e-3)),
    ("box654_poly", "GaussVolPoint", None, None, dict(deltaT=1e-3, mu=1e-3)),
    ("box654_poly", "GaussVolPoint", mixed_box_bcs, None, dict(deltaT=5e-4, mu=1e-3)),
    ("box654", "GaussVolPoint", mixed_box_bcs, None, dict(deltaT=1e-3, mu=2e-3, Pr=0.7, ScQGD=0.8, PrQGD=0.9, alphaQGD=0.4)),
    ("box654_jitter", "GaussVolPoint", mixed_box_bcs, None, dict(deltaT=5e-4, mu=2e-3)),
    ("plane2d", "GaussVolPoint", empty_z_bcs, plane_init, dict(deltaT=1e-3)),
    ("plane2d", "leastSquares", empty_z_bcs, plane_init, dict(deltaT=1e-3)),
    ("plane2d_jitter", "leastSquares", empty_z_bcs, plane_init, dict(deltaT=5e-4, mu=1e-3)),
    ("step2d", "leastSquares", cases.forward_step_bcs, step_init, dict(deltaT=5e-4)),
    ("step2d", "GaussVolPoint", cases.forward_step_bcs, step_init, dict(deltaT=5e-4)),
    ("step2d", "reduced", cases.forward_step_bcs, step_init, dict(deltaT=5e-4)),
]


def _arms_of(mesh_kind, scheme, opt):
    mesh = make_mesh(mesh_kind)
    arms = [a for a, _ in device_pair_arms(mesh, q.default_options(stencil=scheme, **opt))]
    mesh.close()
    return arms


# every fixed-deltaT 3-D GaussVolPoint case runs TWICE against the oracle: through the fused one-launch step -- the kernel the benchmark times --
# and through the separate kernels (what every other branch and qgd_case_update_fluxes run); which one ran is asserted (build_pair)
CASE_ARMS = [c + (arm,) for c in CASES for arm in _arms_of(c[0], c[1], c[4])]


def test_the_case_matrix_covers_the_fused_step():
    assert sum(1 for c in CASE_ARMS if c[5] == "fused") == 7 and sum(1 for c in CASE_ARMS if c[5] == "kernels") == len(CASES)


@pytest.mark.parametrize("mesh_kind,scheme,bc_fn,init_fn,opt,arm", CASE_ARMS)
def test_update_fluxes_and_steps(mesh_kind, scheme, bc_fn, init_fn, opt, arm):
    mesh, dev, gc, oc = build_pair(mesh_kind, scheme, bc_fn, init_fn, arm=arm, **opt)
    tag = (mesh_kind, scheme, arm)
    # state straight after createFields.H
    compare_fields(gc, oc, CELL_FIELDS, 1e-13, tag + ("init",))
    compare_fields(gc, oc, [n + ".boundary" for n in ("rho", "U", "p", "e", "c", "H", "muQGD")], 1e-13, tag + ("init.bnd",))
    # updateFields.H + updateFluxes.H
    gc.updateFluxes()
    oc.updateFluxes()
    compare_fields(gc, oc, FACE_FIELDS, FLUX_TOL, tag + ("fluxes",))
    # explicit steps
    for chunk in (1, 4, 20):
        gc.step(chunk)
        oc.step(chunk)
        compare_fields(gc, oc, ["rho", "U", "p", "e", "rhoU", "rhoE"], STATE_TOL, tag + (f"step+{chunk}",))
    # the reference refreshes H = (rhoE + p)/rho in updateFields.H; the device keeps it current per step
    gc.updateFluxes()
    oc.updateFluxes()
    compare_fields(gc, oc, CELL_FIELDS, STATE_TOL, tag + ("final",))
    compare_fields(gc, oc, FACE_FIELDS, STATE_TOL, tag + ("final fluxes",))
    # min(rho), min(e) [QGDFoam.C L142]: the device monitor covers the steps since the previous query
    gc.info()
    gc.step(1)
    oc.step(1)
    ig, io = gc.info(), oc.info()
    assert ig["steps"] == io["steps"] == 26
    assert abs(ig["time"] - io["time"]) <= 1e-12 * max(1.0, abs(io["time"]))
    assert abs(ig["minRho"] - io["minRho"]) <= 1e-9 and abs(ig["minE"] - io["minE"]) <= 1e-9
    assert ig["minRho"] > 0 and ig["minE"] > 0
    gc.close(); dev.close()


@pytest.mark.parametrize("mesh_kind,scheme,bc_fn,init_fn,arm", [
    ("box654", "GaussVolPoint", None, None, "kernels"),
    ("box654", "GaussVolPoint", None, None, "fusedAdjust"),
    ("box654_poly", "GaussVolPoint", mixed_box_bcs, None, "fusedAdjust"),
    ("step2d", "leastSquares", cases.forward_step_bcs, step_init, "kernels"),
])
def test_adjust_time_step(mesh_kind, scheme, bc_fn, init_fn, arm):
    """QGDCourantNo.H + setDeltaT-QGDQHD.H: Courant number and deltaT follow the oracle -- with the three kernels, and with the cell blocks
    (every block up to its flux sums and Courant partials, then deltaT, then the cell kernel)."""
    mesh, dev, gc, oc = build_pair(mesh_kind, scheme, bc_fn, init_fn, arm=arm, deltaT=1e-4, adjustTimeStep=1, maxCo=0.3,
                                   maxDeltaT=1.0, cTau=0.75)
    for _ in range(10):
        gc.step(1)
        oc.step(1)
        ig, io = gc.info(), oc.info()
        assert abs(ig["deltaT"] - io["deltaT"]) <= 1e-11 * io["deltaT"], (ig, io)
        assert abs(ig["CoNum"] - io["CoNum"]) <= 1e-11 * max(io["CoNum"], 1e-30), (ig, io)
        assert abs(ig["time"] - io["time"]) <= 1e-11 * io["time"]
    compare_fields(gc, oc, ["rho", "U", "p", "e"], STATE_TOL, (mesh_kind, scheme, "adjust"))
    gc.close(); dev.close()


def test_thermo_accessors():
    """QGDThermo accessor surface (QGDThermo.H L99-135) returns the oracle's fields."""
    mesh, dev, gc, oc = build_pair("box654", "GaussVolPoint", None, None, deltaT=1e-3, mu=1e-3)
    gc.step(2); oc.step(2)
    gc.updateFluxes(); oc.updateFluxes()
    th = gc.thermo
    pairs = [(th.tauQGDf(), "tauQGDf"), (th.hQGDf(), "hQGDf"), (th.tauQGD(), "tauQGD"), (th.hQGD(), "hQGD"),
             (th.muQGD(), "muQGD"), (th.alphauQGD(), "alphauQGD"), (th.c(), "c"), (th.p(), "p"), (th.rho(), "rho"),
             (th.mu(), "mu")]
    for got, name in pairs:
        assert rel_err(got, oc.field(name)) <= STATE_TOL, name
    assert th.implicitDiffusion() is False
    gc.close(); dev.close()


@pytest.mark.parametrize("variant", ["hex", "jitter", "jitter+triangles"])
@pytest.mark.parametrize("arm", ["fused", "kernels", "kernels+adjustTimeStep", "fusedAdjust+adjustTimeStep"])
def test_bench_kernel_directly_against_the_oracle(variant, arm):
    """The kernel bench.py times -- fusedFaceCellKernel: a block of <= 128 cells stages its cells and the cells around them in LDS, forms its
    vertex values, computes every internal face of its cells and advances them, ONE launch per step -- against the oracle with nothing in
    between, on a mesh large enough for real blocks (20^3 cells: 64 bricks of 5^3) -- asserted, not assumed: the case says it is
    fused, it has many blocks and none tiny on average, blocks border on blocks (faces on a block's surface are computed by the block
    on either side: facesComputed > nInternalFaces), and no block is the one-block degenerate case of the small parity meshes.  The
    "kernels" arms keep the three kernels it replaced (faceFluxGvp3TileKernel<128> on full face tiles: asserted too) oracle-checked, with a
    fixed deltaT and under Courant-number control.  26 explicit steps."""
    mesh = q.PolyMesh.box(20, 20, 20)
    if variant != "hex":
        mesh.jitter(0.15, seed=2024)
    if variant == "jitter+triangles":
        mesh.split_quads(7)
    adjust = 1 if arm.endswith("adjustTimeStep") else 0
    dev = q.Device(mesh, fused_tables="any" if arm.startswith("fused") else False)
    h = 1.0 / 20
    opt = q.default_options(stencil="GaussVolPoint", deltaT=0.1 * h / 1.3, mu=1e-3, adjustTimeStep=adjust, maxCo=0.25)
    gc = q.QGDFoamCase(dev, opt)
    assert_path(gc, arm.split("+")[0], (variant, arm))
    if arm == "fused":
        fi = gc.fused_info()
        assert fi["blocks"] >= (mesh.nCells + 127) // 128 and fi["blocks"] <= mesh.nCells // 40, fi      # many blocks, none tiny on average
        assert fi["facesComputed"] > mesh.nInternalFaces, fi                                            # rim faces computed on both sides
        assert fi["cellsStaged"] > 2 * mesh.nCells and fi["verticesFormed"] > mesh.nPoints, fi          # the blocks stage their surroundings
        assert 0 < fi["ldsBytes"] <= 64 * 1024, fi
    elif arm.startswith("kernels"):
        ft = dev.face_tiles()
        assert ft["facesPerTile"] == 128 and ft["tiles"] == (mesh.nInternalFaces + 127) // 128, ft
        assert 4 * ft["gatherTiles"] < ft["tiles"], ft
    oc = OracleCase(oracle_mesh_of(mesh), opt)
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    gc.set_fields(U, T, p)
    oc.set_fields(U, T, p)
    gc.step(26)
    oc.step(26)
    compare_fields(gc, oc, ["rho", "U", "p", "e", "rhoE"], STATE_TOL, (variant, arm, "26 steps"))
    ig, io = gc.info(), oc.info()
    assert abs(ig["deltaT"] - io["deltaT"]) <= 1e-11 * io["deltaT"] and abs(ig["time"] - io["time"]) <= 1e-11 * io["time"]
    gc.close(); dev.close()
