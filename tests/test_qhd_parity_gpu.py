"""GPU parity of the QHDFoam flux assembly (qgd_qhd_fluxes) against the oracle, and its analytic properties."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam

import cases
import oracle
from util import make_mesh, oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu

CASES = [("box654_jitter", "GaussVolPoint"), ("box654_tri", "GaussVolPoint"), ("box654", "reduced"),
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
