"""Non-uniform alphaQGD / ScQGD fields [QGDCoeffs.C L119-160, constScPrModel1.C L66-79, L97-131] in the oracle: a uniform
field reproduces the scalar option bit for bit, a non-uniform one enters tauQGDf = lin(alphaQGD/c) hQGDf, tauQGD and muQGD."""
import numpy as np

import qgdsolver_amd as q
import cases
from oracle import OracleCase
from util import make_mesh, oracle_mesh_of


def run(om, mesh, coeffs, steps=3):
    oc = OracleCase(om, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3, alphaQGD=0.4, ScQGD=0.7))
    if coeffs:
        oc.set_qgd_coeffs(**coeffs)
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    oc.set_fields(U, T, p)
    oc.updateFluxes()
    out = {f: oc.field(f) for f in ("tauQGDf", "tauQGD", "muQGD", "phiJm")}
    oc.step(steps)
    out.update({f: oc.field(f) for f in ("rho", "U", "p")})
    return out


def test_uniform_field_equals_scalar_and_nonuniform_enters_the_closure():
    mesh = make_mesh("box654_jitter")
    om = oracle_mesh_of(mesh)
    n, nb, nif = mesh.nCells, mesh.nBoundaryFaces, mesh.nInternalFaces
    base = run(om, mesh, None)
    same = run(om, mesh, dict(alphaQGD=(np.full(n, 0.4), np.full(nb, 0.4)), ScQGD=(np.full(n, 0.7), np.full(nb, 0.7))))
    for f in base:
        assert np.array_equal(base[f], same[f]), f
    rng = np.random.default_rng(1)
    a = 0.3 + 0.3 * rng.random(n)
    sc = 0.5 + rng.random(n)
    own = mesh.array("owner")
    got = run(om, mesh, dict(alphaQGD=(a, a[own[nif:]]), ScQGD=(sc, sc[own[nif:]])))
    # tauQGD = alphaQGD hQGD / c and muQGD = p ScQGD tauQGD cell by cell [constScPrModel1.C L104-111]
    assert np.allclose(got["tauQGD"] / base["tauQGD"], a / 0.4, rtol=1e-13)
    assert np.allclose(got["muQGD"] / base["muQGD"], (a / 0.4) * (sc / 0.7), rtol=1e-13)
    # tauQGDf = lin(alphaQGD/c) hQGDf [L103]
    w, nei = mesh.array("weights")[:nif], mesh.array("neighbour")
    oc = OracleCase(om, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    oc.set_fields(U, T, p)
    c, hf = oc.field("c"), oc.field("hQGDf")
    aoc = a / c
    assert np.allclose(got["tauQGDf"][:nif], (w * (aoc[own[:nif]] - aoc[nei]) + aoc[nei]) * hf[:nif], rtol=1e-13)
    assert np.abs(got["rho"] - base["rho"]).max() > 1e-8


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("kind,stencil", [("box654_jitter", "GaussVolPoint"), ("step2d", "leastSquares")])
def test_device_matches_oracle_with_nonuniform_coefficient_fields(kind, stencil):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    n, nif = mesh.nCells, mesh.nInternalFaces
    rng = np.random.default_rng(2)
    a = 0.3 + 0.3 * rng.random(n)
    sc = 0.5 + rng.random(n)
    own = mesh.array("owner")
    coeffs = dict(alphaQGD=(a, a[own[nif:]]), ScQGD=(sc, sc[own[nif:]]))
    opt = q.default_options(stencil=stencil, deltaT=5e-4, mu=1e-3)
    C = mesh.array("C").reshape(-1, 3)
    if kind == "step2d":
        U = np.zeros((n, 3)); U[:, 0] = 3.0
        T, p = 1.0 + 0.05 * np.sin(2.0 * C[:, 0]), 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])
    else:
        U, T, p = cases.box_initial_fields(C)
    dev = q.Device(mesh)
    gc, oc = q.QGDFoamCase(dev, opt), OracleCase(om, opt)
    for c in (gc, oc):
        if kind == "step2d":
            cases.forward_step_bcs(c)
        c.set_qgd_coeffs(**coeffs)
        c.set_fields(U, T, p)
        c.updateFluxes()
    for f in ("tauQGDf", "phiJm", "phiJmU", "phiPi", "phiQ", "tauQGD", "muQGD", "alphauQGD", "muQGD.boundary"):
        ref = oc.field(f)
        assert np.abs(gc.field(f) - ref).max() <= 1e-12 * np.abs(ref).max(), f
    gc.step(10); oc.step(10)
    for f in ("rho", "U", "p", "e", "muQGD"):
        ref = oc.field(f)
        assert np.abs(gc.field(f) - ref).max() <= 1e-10 * np.abs(ref).max(), f
    gc.close(); dev.close()
