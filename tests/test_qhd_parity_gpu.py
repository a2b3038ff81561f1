"""GPU parity of the QHDFoam flux assembly (qgd_qhd_fluxes) against the oracle, and its analytic properties."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam

import cases
import oracle
from util import make_mesh, oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu

CASES = [("box654_jitter", "GaussVolPoint"), ("box654_tri", "GaussVolPoint"), ("box654_poly", "GaussVolPoint"), ("box654", "reduced"),
         ("plane2d_jitter", "leastSquares"), ("plane2d", "GaussVolPoint"), ("step2d", "leastSquares"), ("line1d", "GaussVolPoint")]


def fields(mesh, seed):
    rng = np.random.default_rng(seed)
    n, nb, nf = mesh.nCells, mesh.nBoundaryFaces, mesh.nFaces
    U = (rng.standard_normal((n, 3)), rng.standard_normal((nb, 3)))
    T = (300 + rng.standard_normal(n), 300 + rng.standard_normal(nb))
    p = (rng.standard_normal(n), rng.standard_normal(nb))
    rho = (1.0 + 0.1 * rng.random(n), 1.0 + 0.1 * rng.random(nb))
    tau = 1e-3 * (1 + rng.random(nf))
    phi = rng.standard_normal(nf)
    return U, T, p, rho, tau, phi


@pytest.mark.parametrize("kind,scheme", CASES)
def test_qhd_fluxes_match_oracle(kind, scheme):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh)
    U, T, p, rho, tau, phi = fields(mesh, 5)
    beta, g = 3.4e-3, (0.0, -9.81, 0.0)
    # first group only (before the pressure equation)
    got1 = qhdfoam.updateFluxes(dev, scheme, U, T, rho, tau, beta, g)
    ref1 = oracle.qhd_fluxes(om, scheme, U, T, rho, tau, beta, g)
    assert sorted(got1) == sorted(ref1) == ["gradTf", "gradUf", "phiTauTReg", "phiu", "phiwo", "taubyrhof"]
    for k in ref1:
        assert rel_err(got1[k], ref1[k]) <= 1e-12, (kind, scheme, k, rel_err(got1[k], ref1[k]))
    # everything
    got = qhdfoam.updateFluxes(dev, scheme, U, T, rho, tau, beta, g, p=p, phi=phi)
    ref = oracle.qhd_fluxes(om, scheme, U, T, rho, tau, beta, g, p=p, phi=phi)
    assert len(ref) == 10
    for k in ref:
        assert rel_err(got[k], ref[k]) <= 1e-12, (kind, scheme, k, rel_err(got[k], ref[k]))
    # phiu = Sf . Uf with linear interpolation, independent of the stencil
    Sf = mesh.array("Sf").reshape(-1, 3)
    own, nei, w = mesh.array("owner"), mesh.array("neighbour"), mesh.array("weights")
    nif = mesh.nInternalFaces
    Uf = w[:nif, None] * (U[0][own[:nif]] - U[0][nei]) + U[0][nei]
    assert np.allclose(got["phiu"][:nif], (Sf[:nif] * Uf).sum(1), rtol=1e-13, atol=1e-15)
    dev.close()


def test_qhd_argument_checks():
    mesh = make_mesh("box654")
    dev = q.Device(mesh)
    U, T, p, rho, tau, phi = fields(mesh, 1)
    with pytest.raises(q.QgdError):
        qhdfoam.updateFluxes(dev, "leastSquares", U, T, rho, tau, 1e-3, (0, 0, -9.81))   # 3-D: refused like fvscOpName
    dev.close()


@pytest.mark.parametrize("kind", ["box654_jitter", "plane2d_jitter"])
def test_qgdInterpolate_qgdFlux_and_tau_closures(kind):
    """QGDInterpolate.H L38-118 helpers and the QHD tau closures (constTau / HbyUQHD / T0byGr / H2bynuQHD)."""
    from qgdsolver_amd import fvsc
    from oracle import OracleCase

    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh)
    own, nei, w = mesh.array("owner"), mesh.array("neighbour"), mesh.array("weights")
    nif = mesh.nInternalFaces
    live = np.ones(mesh.nBoundaryFaces, bool)
    for t, s, z in zip(mesh.array("patchType"), mesh.array("patchStart"), mesh.array("patchSize")):
        if t == q._lib.PATCH_EMPTY:
            live[s - nif: s - nif + z] = False
    for nc in (1, 3, 9):
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, 7)
        got = fvsc.qgdInterpolate(dev, q.volField("f", cell, bnd))
        c2 = cell.reshape(mesh.nCells, -1)
        ref_i = w[:nif, None] * (c2[own[:nif]] - c2[nei]) + c2[nei]
        assert rel_err(got.reshape(mesh.nFaces, -1)[:nif], ref_i) <= 4e-16    # one fma vs mul+add
        assert np.array_equal(got.reshape(mesh.nFaces, -1)[nif:][live], bnd.reshape(mesh.nBoundaryFaces, -1)[live])
    flux = np.random.default_rng(1).standard_normal(mesh.nFaces)
    psif = np.random.default_rng(2).standard_normal((mesh.nFaces, 3))
    assert np.array_equal(fvsc.qgdFlux(dev, flux, psif), flux[:, None] * psif)
    # length scales equal the oracle's QGDCoeffs::updateQGDLength
    oc = OracleCase(om, q.default_options(stencil="reduced"))
    if kind.startswith("plane2d"):
        for patch in (4, 5):
            oc.set_bc(patch, U=("none", None), T=("none", None), p=("none", None))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    oc.set_fields(U, T, p)
    oc.updateFluxes()
    h, hf, hb = fvsc.device_field(dev, "hQGD"), fvsc.device_field(dev, "hQGDf"), fvsc.device_field(dev, "hQGD.boundary")
    assert rel_err(h, oc.field("hQGD")) <= 1e-15 and rel_err(hf, oc.field("hQGDf")) <= 1e-15
    assert rel_err(hb, oc.field("hQGD.boundary")) <= 1e-15
    # closures
    t_const = qhdfoam.tauQGDf(dev, "constTau", Tau=2e-3)
    assert np.array_equal(t_const[:nif], np.full(nif, 2e-3))
    t_hu = qhdfoam.tauQGDf(dev, "HbyUQHD", aQGD=0.4, UQHD=2.0)
    th = 0.4 * h / 2.0
    assert rel_err(t_hu[:nif], w[:nif] * (th[own[:nif]] - th[nei]) + th[nei]) <= 4e-16
    t_gr = qhdfoam.tauQGDf(dev, "T0byGr", T0=3.0, Gr=1.5e3)
    assert np.array_equal(t_gr[:nif], np.full(nif, 3.0 / 1.5e3))
    nu = (1e-3 * (1 + np.random.default_rng(3).random(mesh.nCells)), np.full(mesh.nBoundaryFaces, 1e-3))
    t_h2 = qhdfoam.tauQGDf(dev, "H2bynuQHD", aQGD=0.5, nu=nu)
    th2 = 0.5 * h * h / nu[0]
    assert rel_err(t_h2[:nif], w[:nif] * (th2[own[:nif]] - th2[nei]) + th2[nei]) <= 4e-16
    dev.close()
