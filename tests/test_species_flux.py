"""Species flux block (reactingLagrangianQGDFoam/updateFluxes.H L117-132): oracle properties on CPU, device parity on GPU."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qgdfoam
import oracle as orc
from test_qhd_pressure import HostDev
from util import make_mesh, oracle_mesh_of, rel_err


def inputs(mesh, seed, linear=False):
    rng = np.random.default_rng(seed)
    C = mesh.array("C").reshape(-1, 3)
    Cf = mesh.array("Cf").reshape(-1, 3)[mesh.nInternalFaces:]
    if linear:
        coef = np.array([0.3, -0.7, 0.2])
        Y = (0.5 + C @ coef, 0.5 + Cf @ coef)
    else:
        Y = (rng.random(mesh.nCells), rng.random(mesh.nBoundaryFaces))
    U = (rng.standard_normal((mesh.nCells, 3)), rng.standard_normal((mesh.nBoundaryFaces, 3)))
    phiJm = rng.standard_normal(mesh.nFaces)
    phi = rng.standard_normal(mesh.nFaces)
    tau = 1e-3 * (1 + rng.random(mesh.nFaces))
    return Y, U, phiJm, phi, tau


def oracle_call(om):
    def call(*a):
        assert orc.species_flux(om, *a) == 0
    return call


def test_oracle_linear_species_field():
    """for Y linear in space every fvsc stencil returns the exact gradient on an orthogonal box, so the regularising flux
    is -phi*tau*(Uf . grad Y) exactly and phiJmY = phiJm*Yf + that"""
    mesh = q.PolyMesh.box(7, 6, 5)
    om = oracle_mesh_of(mesh)
    Y, U, phiJm, phi, tau = inputs(mesh, 1, linear=True)
    # faces none of whose vertices lies on the boundary (there the vertex values are one-sided averages)
    fo, fp = mesh.array("faceOffsets"), mesh.array("facePoints")
    on_boundary = np.zeros(mesh.nPoints, dtype=bool)
    on_boundary[fp[fo[mesh.nInternalFaces]:]] = True
    inner = np.array([not on_boundary[fp[fo[f]:fo[f + 1]]].any() for f in range(mesh.nInternalFaces)])
    assert inner.sum() > 20
    coef = np.array([0.3, -0.7, 0.2])
    nIF = mesh.nInternalFaces
    w = mesh.array("weights")[:nIF]
    own, nei = mesh.array("owner"), mesh.array("neighbour")
    Uf = np.vstack([w[:, None] * U[0][own[:nIF]] + (1 - w)[:, None] * U[0][nei], U[1]])
    Yf = np.concatenate([w * Y[0][own[:nIF]] + (1 - w) * Y[0][nei], Y[1]])
    for scheme in ("GaussVolPoint", "reduced"):
        r = qgdfoam.speciesFlux(HostDev(mesh), scheme, Y, U, phiJm, phi, tau, call=oracle_call(om))
        if scheme == "GaussVolPoint":
            assert np.abs(r["gradYf"][:nIF][inner] - coef).max() <= 1e-12
            want = -phi * tau * (Uf @ coef)
            assert np.abs(r["diffusiveFlux"][:nIF][inner] - want[:nIF][inner]).max() <= 1e-14
        assert np.abs(r["phiJmY"] - (phiJm * Yf + r["diffusiveFlux"])).max() <= 1e-15


def test_oracle_refuses_leastsquares_in_3d():
    mesh = q.PolyMesh.box(3, 3, 3)
    om = oracle_mesh_of(mesh)
    Y, U, phiJm, phi, tau = inputs(mesh, 2)
    z = np.zeros(mesh.nFaces)
    assert orc.species_flux(om, "leastSquares", Y[0], Y[1], U[0].reshape(-1), U[1].reshape(-1), phiJm, phi, tau, z, z.copy(),
                            np.zeros(3 * mesh.nFaces)) == -4


@pytest.mark.gpu
@pytest.mark.parametrize("kind,scheme", [("box654_poly", "GaussVolPoint"), ("box654_jitter", "reduced"), ("plane2d_jitter", "leastSquares"),
                                         ("plane2d_jitter", "GaussVolPoint"), ("step2d", "GaussVolPoint"), ("line1d", "GaussVolPoint")])
def test_device_matches_oracle(kind, scheme):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    Y, U, phiJm, phi, tau = inputs(mesh, 7)
    ref = qgdfoam.speciesFlux(HostDev(mesh), scheme, Y, U, phiJm, phi, tau, call=oracle_call(om))
    dev = q.Device(mesh)
    got = qgdfoam.speciesFlux(dev, scheme, Y, U, phiJm, phi, tau)
    for k in ref:
        assert rel_err(got[k], ref[k]) <= 1e-12, (kind, scheme, k)
    if mesh.nGeometricD == 3:
        with pytest.raises(q.QgdError):
            qgdfoam.speciesFlux(dev, "leastSquares", Y, U, phiJm, phi, tau)
    dev.close()


# ---- the species equation itself (QGDYEqn.H L67-86, explicit branch): qgd_species_step -------------------------------------------
def step_inputs(mesh, seed):
    rng = np.random.default_rng(seed)
    Y = (0.2 + 0.6 * rng.random(mesh.nCells), 0.2 + 0.6 * rng.random(mesh.nBoundaryFaces))
    rho_old = 1.0 + 0.1 * rng.random(mesh.nCells)
    rho = rho_old * (1.0 + 0.01 * rng.standard_normal(mesh.nCells))
    phiJmY = 1e-3 * rng.standard_normal(mesh.nFaces)
    muf = 1e-3 * (1.0 + rng.random(mesh.nFaces))
    Su = 0.05 * rng.standard_normal(mesh.nCells)
    return Y, rho_old, rho, phiJmY, muf, Su


def test_oracle_species_step_conserves_the_species_mass():
    """sum V rho Y changes by -dt (boundary fluxes of phiJmY - laplacian flux) + dt sum V Su: the internal faces cancel in pairs"""
    mesh = make_mesh("box654_poly")
    om = oracle_mesh_of(mesh)
    Y, rho_old, rho, phiJmY, muf, Su = step_inputs(mesh, 4)
    dt, Sc = 1e-3, 0.8
    df = np.zeros(mesh.nFaces)

    def call(*a):
        assert orc.species_step(om, *a) == 0
    new = qgdfoam.speciesStep(HostDev(mesh), Y, rho_old, rho, phiJmY, muf, Sc, dt, df, Su=Su, call=call)
    assert new.min() > 0          # nothing clipped with these inputs
    V, nif = mesh.array("V"), mesh.nInternalFaces
    types = mesh.array("patchType")
    live = np.ones(mesh.nBoundaryFaces, dtype=bool)
    for ip in range(mesh.nPatches):
        if types[ip] == q._lib.PATCH_EMPTY:
            s0 = mesh.array("patchStart")[ip] - nif
            live[s0:s0 + mesh.array("patchSize")[ip]] = False
    lhs = (V * rho * new).sum() - (V * rho_old * Y[0]).sum()
    rhs = -dt * ((phiJmY[nif:] - df[nif:])[live]).sum() + dt * (V * Su).sum()
    assert abs(lhs - rhs) <= 1e-13 * (V * rho_old * Y[0]).sum()
    # the laplacian flux added to diffusiveFlux is (muf/Sc) snGrad(Y) |Sf| with the uncorrected snGrad
    own, nei = mesh.array("owner"), mesh.array("neighbour")
    want = muf[:nif] / Sc * mesh.array("nonOrthDeltaCoeffs")[:nif] * (Y[0][nei] - Y[0][own[:nif]]) * mesh.array("magSf")[:nif]
    assert np.abs(df[:nif] - want).max() <= 1e-15 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["box654_poly", "box654_jitter", "plane2d_jitter", "step2d"])
def test_device_species_step_matches_oracle(kind):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh)
    Y, rho_old, rho, phiJmY, muf, Su = step_inputs(mesh, 6)
    Y[0][:3] = 1e-7            # cells the step drives below zero: Yi.max(0.0)
    phiJmY[:mesh.nInternalFaces] += 0.0
    dt, Sc = 2e-3, 1.3
    for su in (Su, None):
        dfo, dfd = 0.01 * np.ones(mesh.nFaces), 0.01 * np.ones(mesh.nFaces)

        def call(*a):
            assert orc.species_step(om, *a) == 0
        want = qgdfoam.speciesStep(HostDev(mesh), Y, rho_old, rho, phiJmY, muf, Sc, dt, dfo, Su=su, call=call)
        got = qgdfoam.speciesStep(dev, Y, rho_old, rho, phiJmY, muf, Sc, dt, dfd, Su=su)
        assert rel_err(got, want) <= 1e-12 and rel_err(dfd, dfo) <= 1e-13, (kind, rel_err(got, want))
    dev.close()


# ---- the implicitDiffusion branch of the species equation [QGDYEqn.H L47-66] (VERDICT r03 "missing" #4) ---------------------------------
def fixed_faces(mesh, patches=(0,)):
    nif = mesh.nInternalFaces
    fx = np.zeros(mesh.nBoundaryFaces, dtype=np.uint8)
    for ip in patches:
        s0 = mesh.array("patchStart")[ip] - nif
        fx[s0:s0 + mesh.array("patchSize")[ip]] = 1
    return fx


def test_oracle_implicit_species_step():
    """mass: sum V rho Y changes by -dt (boundary phiJmY + boundary YEqn.flux()) + dt sum V Su; diffusiveFlux gets YEqn.flux() = -a (Y_N - Y_O)
    of the NEW Y (the sign "- fvm::laplacian" gives it); as deltaT -> 0 the two branches meet; without diffusion they are one"""
    mesh = make_mesh("box654_poly")
    om = oracle_mesh_of(mesh)
    Y, rho_old, rho, phiJmY, muf, Su = step_inputs(mesh, 4)
    dt, Sc = 1e-3, 0.8
    fx = fixed_faces(mesh)
    df = np.zeros(mesh.nFaces)

    def call(*a):
        assert orc.species_step_implicit(om, *a) == 0
    new, info = qgdfoam.speciesStepImplicit(HostDev(mesh), Y, rho_old, rho, phiJmY, muf, Sc, dt, df, Su=Su, fixedValueFaces=fx, tolerance=1e-15,
                                            maxIter=500, call=call)
    assert new.min() > 0 and 0 < info["iterations"] < 100 and info["final"] < 1e-14 < info["initial"]
    V, nif = mesh.array("V"), mesh.nInternalFaces
    own, nei = mesh.array("owner"), mesh.array("neighbour")
    a = muf / Sc * mesh.array("magSf") * np.where(np.arange(mesh.nFaces) < nif, mesh.array("nonOrthDeltaCoeffs"), mesh.array("deltaCoeffs"))
    want = -a[:nif] * (new[nei] - new[own[:nif]])
    assert np.abs(df[:nif] - want).max() <= 1e-13 * np.abs(want).max()
    wb = np.where(fx == 1, -a[nif:] * (Y[1] - new[own[nif:]]), 0.0)
    assert np.abs(df[nif:] - wb).max() <= 1e-13 * max(np.abs(wb).max(), 1e-300)
    types = mesh.array("patchType")
    live = np.ones(mesh.nBoundaryFaces, dtype=bool)
    for ip in range(mesh.nPatches):
        if types[ip] == q._lib.PATCH_EMPTY:
            s0 = mesh.array("patchStart")[ip] - nif
            live[s0:s0 + mesh.array("patchSize")[ip]] = False
    lhs = (V * rho * new).sum() - (V * rho_old * Y[0]).sum()
    rhs = -dt * ((phiJmY[nif:] + df[nif:])[live]).sum() + dt * (V * Su).sum()
    assert abs(lhs - rhs) <= 1e-12 * (V * rho_old * Y[0]).sum()
    # no diffusion: the explicit branch
    z = np.zeros(mesh.nFaces)
    new0, _ = qgdfoam.speciesStepImplicit(HostDev(mesh), Y, rho_old, rho, phiJmY, 0.0 * muf, Sc, dt, z.copy(), Su=Su, fixedValueFaces=fx, tolerance=1e-15, call=call)
    ex0 = qgdfoam.speciesStep(HostDev(mesh), Y, rho_old, rho, phiJmY, 0.0 * muf, Sc, dt, z.copy(), Su=Su, call=lambda *b: orc.species_step(om, *b))
    assert rel_err(new0, ex0) <= 1e-14
    # first order in deltaT between the branches (all patches fixedValue here: the explicit step treats every patch value as one)
    allfx = np.ones(mesh.nBoundaryFaces, dtype=np.uint8)
    diffs = []
    for k in (1, 2):
        h = 2e-4 / k
        im, _ = qgdfoam.speciesStepImplicit(HostDev(mesh), Y, rho_old, rho_old, 0 * phiJmY, muf, Sc, h, z.copy(), fixedValueFaces=allfx, tolerance=1e-15, call=call)
        ex = qgdfoam.speciesStep(HostDev(mesh), Y, rho_old, rho_old, 0 * phiJmY, muf, Sc, h, z.copy(), call=lambda *b: orc.species_step(om, *b))
        diffs.append(np.abs(im - ex).max())
    assert 0 < diffs[1] < 0.3 * diffs[0], diffs          # the difference of one step is O(deltaT^2)


def test_species_implicit_step_from_the_listing_text():
    """tests/golden/ref_expr_specieseqn_implicit.npz: QGDYEqn.H L40-66, L86-92 executed as listed on the two-cell mesh (three species, the
    last one inert): YEqn.solve(), diffusiveFlux[i] += YEqn.flux(), the inert species' bookkeeping"""
    import ref_expr_cases as rc
    from test_ref_expr import oracle_mesh
    g = rc.load("specieseqn_implicit")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        mesh = type("M", (), {"nCells": 2, "nFaces": om.nFaces, "nBoundaryFaces": om.nBoundaryFaces})()
        dev = type("D", (), {"mesh": mesh})()
        nf = om.nFaces
        ns, inert = g["Y"][i].shape[0], int(g["inertIndex"][i])
        Y = [(g["Y"][i][k].copy(), np.zeros(om.nBoundaryFaces)) for k in range(ns)]
        jm = [np.concatenate([[g["phiJmY"][i][k]], np.zeros(nf - 1)]) for k in range(ns)]
        muf = np.full(nf, float(g["muf"][i]))
        df = [np.concatenate([[g["diffusiveFlux0"][i][k]], np.zeros(nf - 1)]) for k in range(ns)]
        new = qgdfoam.QGDYEqn(dev, Y, g["rhoOld"][i], g["rho"][i], jm, muf, list(g["Sc"][i]), float(g["deltaT"][i]), df, inert, Su=list(g["Su"][i]),
                              implicitDiffusion=True, tolerance=1e-15, maxIter=50, call=lambda *a: orc.species_step_implicit(om, *a))
        for k in range(ns):
            assert rel_err(new[k], g["Ynew"][i][k]) <= 1e-11, (i, k, new[k], g["Ynew"][i][k])
            assert abs(df[k][0] - g["diffusiveFlux1"][i][k]) <= 1e-11 * max(np.abs(g["diffusiveFlux1"][i]).max(), 1e-300), (i, k)
        om.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["box654_poly", "box654_jitter", "plane2d_jitter", "step2d"])
def test_device_implicit_species_step_matches_oracle(kind):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh)
    Y, rho_old, rho, phiJmY, muf, Su = step_inputs(mesh, 6)
    Y[0][:3] = 1e-7
    dt, Sc = 2e-3, 1.3
    fx = fixed_faces(mesh, (0, 1))
    if mesh.nGeometricD == 2:
        types = mesh.array("patchType")
        for ip in range(mesh.nPatches):
            if types[ip] == q._lib.PATCH_EMPTY:
                s0 = mesh.array("patchStart")[ip] - mesh.nInternalFaces
                fx[s0:s0 + mesh.array("patchSize")[ip]] = 0
    for su, fixed in ((Su, fx), (None, None)):
        dfo, dfd = 0.01 * np.ones(mesh.nFaces), 0.01 * np.ones(mesh.nFaces)
        want, io = qgdfoam.speciesStepImplicit(HostDev(mesh), Y, rho_old, rho, phiJmY, muf, Sc, dt, dfo, Su=su, fixedValueFaces=fixed, tolerance=1e-14,
                                               maxIter=500, call=lambda *a: orc.species_step_implicit(om, *a))
        got, ig = qgdfoam.speciesStepImplicit(dev, Y, rho_old, rho, phiJmY, muf, Sc, dt, dfd, Su=su, fixedValueFaces=fixed, tolerance=1e-14, maxIter=500)
        assert rel_err(got, want) <= 1e-11 and rel_err(dfd, dfo) <= 1e-10, (kind, rel_err(got, want), rel_err(dfd, dfo))
        assert 0 < ig["iterations"] < 500 and ig["final"] < 1e-13 and abs(ig["initial"] - io["initial"]) <= 1e-9 * io["initial"], (ig, io)
    dev.close()
