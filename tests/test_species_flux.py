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
