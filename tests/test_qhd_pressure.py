"""QHDFoam's pressure equation (QHDpEqn.H L35-47): the oracle against an independent sparse direct solve and against the
properties the equation exists for (the corrected flux phi is divergence free; linear pressure fields are exact)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
from qgdsolver_amd import qhdfoam
import oracle as orc
from util import make_mesh, oracle_mesh_of


class HostDev:
    """what qhdfoam.pEqn needs of a device when the oracle does the work"""

    def __init__(self, mesh):
        self.mesh = mesh


def synthetic(mesh, seed):
    rng = np.random.default_rng(seed)
    nF, nB = mesh.nFaces, mesh.nBoundaryFaces
    phiu = 1e-2 * rng.standard_normal(nF)
    phiwo = 1e-3 * rng.standard_normal(nF)
    tbr = 1e-3 * (1.0 + 0.3 * rng.random(nF))
    pb = 1.0 + 0.1 * rng.standard_normal(nB)
    gb = 0.1 * rng.standard_normal(nB)
    return phiu, phiwo, tbr, pb, gb


def divergence(mesh, phi):
    own, nei, nIF = mesh.array("owner"), mesh.array("neighbour"), mesh.nInternalFaces
    d = np.zeros(mesh.nCells)
    np.add.at(d, own, phi)
    np.subtract.at(d, nei, phi[:nIF])
    return d


def direct_solve(mesh, phiu, phiwo, tbr, kinds, pb, gb, ref_cell, ref_value):
    """independent restatement with scipy: assemble A p = b and solve it directly"""
    own, nei, nIF, nC = mesh.array("owner"), mesh.array("neighbour"), mesh.nInternalFaces, mesh.nCells
    magSf, dn = mesh.array("magSf"), mesh.array("nonOrthDeltaCoeffs").copy()
    dn[nIF:] = mesh.array("deltaCoeffs")[nIF:]
    a = tbr * magSf * dn
    ps, pz, pt = mesh.array("patchStart"), mesh.array("patchSize"), mesh.array("patchType")
    live = np.ones(mesh.nFaces, dtype=bool)
    kind_f = np.zeros(mesh.nFaces, dtype=int)
    for i in range(mesh.nPatches):
        sl = slice(ps[i], ps[i] + pz[i])
        if pt[i] == L.PATCH_EMPTY:
            live[sl] = False
        if pt[i] == L.PATCH_GENERIC:
            kind_f[sl] = {"fixedValue": 1, "fixedGradient": 2, "qhdFlux": 2}.get(kinds[i], 0)
    flux = np.where(live, phiu - phiwo, 0.0)
    b = -divergence(mesh, flux)
    A = sp.lil_matrix((nC, nC))
    for f in range(nIF):
        o, n = own[f], nei[f]
        A[o, o] += a[f]; A[n, n] += a[f]; A[o, n] -= a[f]; A[n, o] -= a[f]
    for f in range(nIF, mesh.nFaces):
        o, bi = own[f], f - nIF
        if kind_f[f] == 1:
            A[o, o] += a[f]; b[o] += a[f] * pb[bi]
        elif kind_f[f] == 2:
            b[o] += tbr[f] * magSf[f] * gb[bi]
    if not np.any(kind_f == 1) and ref_cell >= 0:
        b[ref_cell] += A[ref_cell, ref_cell] * ref_value
        A[ref_cell, ref_cell] *= 2
    p = spla.spsolve(A.tocsr(), b)
    phi = np.where(live, phiu - phiwo, 0.0)
    phi[:nIF] += -a[:nIF] * (p[nei] - p[own[:nIF]])
    fb = np.arange(nIF, mesh.nFaces)
    phi[fb] += np.where(kind_f[fb] == 1, -a[fb] * (pb - p[own[fb]]), np.where(kind_f[fb] == 2, -tbr[fb] * magSf[fb] * gb, 0.0))
    return p, np.where(live, phi, 0.0)


CASES = [
    ("box654_jitter", ["fixedValue", "zeroGradient", "qhdFlux", "zeroGradient", "fixedValue", "zeroGradient"]),
    ("box654_poly", ["zeroGradient", "fixedValue", "zeroGradient", "zeroGradient", "fixedGradient", "zeroGradient"]),
    ("box654_tri", ["zeroGradient"] * 6),                                   # all Neumann: the reference level fixes p
    ("plane2d_jitter", ["fixedValue", "zeroGradient", "qhdFlux", "zeroGradient", "none", "none"]),
    ("step2d", ["fixedValue", "zeroGradient", "qhdFlux", "qhdFlux", "qhdFlux", "none"]),
]


@pytest.mark.parametrize("kind,kinds", CASES)
def test_oracle_against_a_direct_solve(kind, kinds):
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    phiu, phiwo, tbr, pb, gb = synthetic(mesh, 3)
    neumann = "fixedValue" not in kinds
    if neumann:  # a pure Neumann problem needs compatible data: make the boundary flux balance the interior sources
        phiu[mesh.nInternalFaces:] = 0.0
        phiwo[mesh.nInternalFaces:] = 0.0
    call = lambda *a: orc.qhd_pressure(om, *a)  # noqa: E731
    p, phi, info = qhdfoam.pEqn(HostDev(mesh), phiu, phiwo, tbr, np.ones(mesh.nCells), kinds, pb, gb, tolerance=1e-14, maxIter=5000,
                                pRefCell=7, pRefValue=1.25, call=call)
    assert info["finalResidual"] < 1e-13 and 0 < info["iterations"] < 5000
    pd, phid = direct_solve(mesh, phiu, phiwo, tbr, kinds, pb, gb, 7, 1.25)
    assert np.abs(p - pd).max() <= 1e-9 * max(np.abs(pd).max(), 1.0), kind
    assert np.abs(phi - phid).max() <= 1e-9 * np.abs(phid).max(), kind
    # what the pressure equation is for: the corrected volumetric flux is divergence free
    scale = np.abs(phi).max()
    d = divergence(mesh, phi)
    if neumann:  # the doubled diagonal of the reference cell leaves its own residual: a (p_ref - value) diag term
        d[7] = 0.0
    assert np.abs(d).max() <= 1e-10 * scale, kind


def test_linear_pressure_is_reproduced_exactly():
    """Gamma = 1, no sources: p = 2 - x between fixedValue patches is the discrete solution on an orthogonal box"""
    mesh = q.PolyMesh.box(8, 3, 2)
    om = oracle_mesh_of(mesh)
    nF, nB = mesh.nFaces, mesh.nBoundaryFaces
    pb = np.zeros(nB)
    ps, pz = mesh.array("patchStart"), mesh.array("patchSize")
    pb[ps[0] - mesh.nInternalFaces: ps[0] - mesh.nInternalFaces + pz[0]] = 2.0
    pb[ps[1] - mesh.nInternalFaces: ps[1] - mesh.nInternalFaces + pz[1]] = 1.0
    kinds = ["fixedValue", "fixedValue"] + ["zeroGradient"] * 4
    p, phi, info = qhdfoam.pEqn(HostDev(mesh), np.zeros(nF), np.zeros(nF), np.ones(nF), np.zeros(mesh.nCells), kinds, pb, None,
                                tolerance=1e-15, maxIter=500, call=lambda *a: orc.qhd_pressure(om, *a))
    x = mesh.array("C").reshape(-1, 3)[:, 0]
    assert np.abs(p - (2.0 - x)).max() <= 1e-12
    # flux = -Gamma |S| dp/dn = +|S| through every x-face, nothing through the others
    Sf = mesh.array("Sf").reshape(-1, 3)
    assert np.abs(phi - Sf[:, 0]).max() <= 1e-12


def test_reference_level_is_ignored_when_a_patch_fixes_the_value():
    mesh = make_mesh("box654_jitter")
    om = oracle_mesh_of(mesh)
    phiu, phiwo, tbr, pb, gb = synthetic(mesh, 5)
    kinds = ["fixedValue"] + ["zeroGradient"] * 5
    run = lambda ref: qhdfoam.pEqn(HostDev(mesh), phiu, phiwo, tbr, np.zeros(mesh.nCells), kinds, pb, gb, tolerance=1e-13,  # noqa: E731
                                   maxIter=3000, pRefCell=ref, pRefValue=9.0, call=lambda *a: orc.qhd_pressure(om, *a))[0]
    assert np.array_equal(run(3), run(-1))


def test_qhdflux_gradient_formula():
    g = qhdfoam.qhdFluxGradient(np.array([2.0, -1.0]), np.array([0.5, 0.25]), np.array([1.2, 1.0]), np.array([0.1, 0.2]))
    assert np.allclose(g, [-(2.0 / 0.5 * 1.2 / 0.1), -(-1.0 / 0.25 * 1.0 / 0.2)])
