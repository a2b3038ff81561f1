"""CPU: known-answer properties that pin the oracle.  The reference ships no tests or golden
vectors (PARITY UNPINNED), so the oracle is held to what the listings themselves imply."""
import numpy as np
import pytest

import qgdsolver_amd as q

import cases
from oracle import OracleCase
from util import make_mesh, oracle_mesh_of


def nf_of(om):
    Sf = om.array("Sf").reshape(-1, 3)
    return Sf / np.maximum(om.array("magSf"), 1e-300)[:, None]


def sngrad(mesh, om, cell, bnd):
    """nonOrthDeltaCoeffs*(phi_N - phi_O) on internal faces, deltaCoeffs*(phi_b - phi_O) on patches"""
    own, nei = mesh.array("owner"), mesh.array("neighbour")
    nif = mesh.nInternalFaces
    cell = cell.reshape(mesh.nCells, -1)
    bnd = bnd.reshape(mesh.nBoundaryFaces, -1)
    out = np.zeros((mesh.nFaces, cell.shape[1]))
    out[:nif] = om.array("nonOrthDeltaCoeffs")[:nif, None] * (cell[nei] - cell[own[:nif]])
    out[nif:] = om.array("deltaCoeffs")[nif:, None] * (bnd - cell[own[nif:]])
    return out


def interior_faces(mesh):
    """internal faces none of whose vertices lies on the boundary"""
    fo, fp = mesh.array("faceOffsets"), mesh.array("facePoints")
    onb = np.zeros(mesh.nPoints, bool)
    onb[fp[fo[mesh.nInternalFaces]:]] = True
    return np.array([f for f in range(mesh.nInternalFaces) if not onb[fp[fo[f]:fo[f + 1]]].any()])


@pytest.mark.parametrize("kind", ["box654_jitter", "plane2d", "line1d"])
def test_reduced_is_nf_times_sngrad(kind):
    """Property 1 [reducedFaceNormalStencil.C:71,85,92,105]"""
    mesh = make_mesh(kind); om = oracle_mesh_of(mesh)
    nf = nf_of(om)
    live = np.ones(mesh.nFaces, bool)
    pt, ps, pz = mesh.array("patchType"), mesh.array("patchStart"), mesh.array("patchSize")
    for t, s, z in zip(pt, ps, pz):
        if t == q._lib.PATCH_EMPTY:
            live[s:s + z] = False
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 3)
    sn = sngrad(mesh, om, cell, bnd)
    rc, g = om.fvsc("reduced", "grad_s", cell, bnd)
    assert rc == 0 and np.array_equal(g[live], (nf * sn)[live])
    cellv, bndv = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 3, 4)
    snv = sngrad(mesh, om, cellv, bndv)
    rc, gv = om.fvsc("reduced", "grad_v", cellv, bndv)
    assert np.array_equal(gv[live], (nf[:, :, None] * snv[:, None, :]).reshape(-1, 9)[live])
    rc, dv = om.fvsc("reduced", "div_v", cellv, bndv)
    ref = nf[:, 0] * snv[:, 0] + nf[:, 1] * snv[:, 1] + nf[:, 2] * snv[:, 2]
    assert np.array_equal(dv[live], ref[live])


def test_gaussvolpoint_1d_equals_reduced():
    """Property 2 [GaussVolPointBase1D.C:53-77]"""
    mesh = make_mesh("line1d"); om = oracle_mesh_of(mesh)
    for op, nc in (("grad_s", 1), ("grad_v", 3), ("div_v", 3), ("div_t", 9)):
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, 5)
        a = om.fvsc("GaussVolPoint", op, cell, bnd)[1]
        b = om.fvsc("reduced", op, cell, bnd)[1]
        assert np.array_equal(a, b), op


def test_gaussvolpoint_polygon_faces_fall_back_to_reduced():
    """Property 2, second half [GaussVolPointBase3D.C:759-768, 808-817]: faces with more than 4 vertices take nf*snGrad"""
    mesh = make_mesh("box654_poly"); om = oracle_mesh_of(mesh)
    sizes = np.diff(mesh.array("faceOffsets"))
    poly = np.where(sizes > 4)[0]
    assert len(poly) > 50 and (poly >= mesh.nInternalFaces).any() and (poly < mesh.nInternalFaces).any()
    for op, nc in (("grad_s", 1), ("grad_v", 3), ("div_v", 3), ("div_t", 9)):
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, 11)
        a = om.fvsc("GaussVolPoint", op, cell, bnd)[1]
        b = om.fvsc("reduced", op, cell, bnd)[1]
        assert np.array_equal(a[poly], b[poly]), op
        quad = np.where(sizes[:mesh.nInternalFaces] == 4)[0]
        assert not np.array_equal(a[quad], b[quad])


@pytest.mark.parametrize("kind,scheme", [("plane2d", "leastSquares"), ("plane2d", "GaussVolPoint"), ("plane2d_y", "leastSquares"),
                                         ("plane2d_y", "GaussVolPoint")])
def test_linear_field_exact_2d(kind, scheme):
    """Property 3: a linear field's gradient is reproduced on interior faces of a uniform mesh
    [CalcW.C:70-78,143-150] [ScalarGrad.C:67-70] [2D.C:317-328]"""
    mesh = q.PolyMesh.box(10, 9, 1, hi=(1.0, 0.9, 0.1), patch_types=[0, 0, 0, 0, 1, 1]) if kind == "plane2d" else \
        q.PolyMesh.box(10, 1, 9, hi=(1.0, 0.1, 0.9), patch_types=[0, 0, 1, 1, 0, 0])
    om = oracle_mesh_of(mesh)
    C = om.array("C").reshape(-1, 3); Cf = om.array("Cf").reshape(-1, 3)
    g = np.array([1.3, -0.7, 0.0]) if kind == "plane2d" else np.array([1.3, 0.0, 0.4])
    cell = C @ g + 2.0
    bnd = Cf[mesh.nInternalFaces:] @ g + 2.0
    rc, out = om.fvsc(scheme, "grad_s", cell, bnd)
    assert rc == 0
    # faces away from the real patches (the empty planes touch every point in 2-D, so select by position)
    fc = Cf[:mesh.nInternalFaces]
    inplane = [0, 1] if kind == "plane2d" else [0, 2]
    ext = [1.0, 0.9]
    sel = np.ones(mesh.nInternalFaces, bool)
    for d, e in zip(inplane, ext):
        sel &= (fc[:, d] > 0.15 * e) & (fc[:, d] < 0.85 * e)
    assert sel.sum() > 20
    assert np.abs(out[:mesh.nInternalFaces][sel] - g).max() < 1e-12


def test_linear_field_exact_gaussvolpoint_3d():
    mesh = q.PolyMesh.box(8, 8, 8); om = oracle_mesh_of(mesh)
    C = om.array("C").reshape(-1, 3); Cf = om.array("Cf").reshape(-1, 3)
    g = np.array([1.3, -0.7, 0.4])
    rc, out = om.fvsc("GaussVolPoint", "grad_s", C @ g + 2.0, Cf[mesh.nInternalFaces:] @ g + 2.0)
    idx = interior_faces(mesh)
    assert len(idx) > 500 and np.abs(out[idx] - g).max() < 1e-13
    # vector field U_j = A[j] . x  ->  out[3 i + j] = d_i U_j = A[j][i]     (property 5: tensor layout)
    A = np.array([[1.0, 2.0, 3.0], [-4.0, 5.0, 6.0], [7.0, -8.0, 9.0]])
    rc, gv = om.fvsc("GaussVolPoint", "grad_v", C @ A.T, Cf[mesh.nInternalFaces:] @ A.T)
    assert np.abs(gv[idx].reshape(-1, 3, 3) - A.T).max() < 1e-12
    rc, dv = om.fvsc("GaussVolPoint", "div_v", C @ A.T, Cf[mesh.nInternalFaces:] @ A.T)
    assert np.abs(dv[idx] - np.trace(A)).max() < 1e-12


def test_tensor_layout_leastsquares():
    """Property 5 [leastSquaresStencil.C:155-165]"""
    mesh = q.PolyMesh.box(10, 9, 1, hi=(1.0, 0.9, 0.1), patch_types=[0, 0, 0, 0, 1, 1]); om = oracle_mesh_of(mesh)
    C = om.array("C").reshape(-1, 3); Cf = om.array("Cf").reshape(-1, 3)
    A = np.array([[1.0, 2.0, 0.0], [-4.0, 5.0, 0.0], [7.0, -8.0, 0.0]])
    rc, gv = om.fvsc("leastSquares", "grad_v", C @ A.T, Cf[mesh.nInternalFaces:] @ A.T)
    fc = Cf[:mesh.nInternalFaces]
    sel = (fc[:, 0] > 0.15) & (fc[:, 0] < 0.85) & (fc[:, 1] > 0.15) & (fc[:, 1] < 0.75)
    assert np.abs(gv[:mesh.nInternalFaces][sel].reshape(-1, 3, 3) - A.T).max() < 1e-12


def test_leastsquares_constraint_patches_stay_zero():
    """Quirk B4 [ScalarGrad.C:90-101]: symmetryPlane/empty patches keep a zero gradient, generic patches get nf*snGrad"""
    mesh = make_mesh("box_sym"); om = oracle_mesh_of(mesh)
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 9)
    rc, g = om.fvsc("leastSquares", "grad_s", cell, bnd)
    ps, pz = mesh.array("patchStart"), mesh.array("patchSize")
    assert np.abs(g[ps[2]:ps[2] + pz[2]]).max() == 0.0          # symmetryPlane
    assert np.abs(g[ps[4]:ps[5] + pz[5]]).max() == 0.0          # empty
    assert np.abs(g[ps[0]:ps[0] + pz[0]]).max() > 0.0
    nf = nf_of(om); sn = sngrad(mesh, om, cell, bnd)
    assert np.array_equal(g[ps[0]:ps[0] + pz[0]], (nf * sn)[ps[0]:ps[0] + pz[0]])


def test_triangle_vector_gradient_pattern():
    """Quirk B2 [3D.C:844-854]: on interior triangles every row of grad(U) holds (d_x U_x, d_y U_y, d_z U_z)."""
    mesh = make_mesh("box654_tri"); om = oracle_mesh_of(mesh)
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 3, 2)
    rc, gv = om.fvsc("GaussVolPoint", "grad_v", cell, bnd)
    fo = mesh.array("faceOffsets")
    tri = np.where(np.diff(fo)[:mesh.nInternalFaces] == 3)[0]
    assert len(tri) > 10
    g = gv[tri].reshape(-1, 3, 3)
    assert np.array_equal(g[:, 0, :], g[:, 1, :]) and np.array_equal(g[:, 0, :], g[:, 2, :])
    # the scalar gradients of the components give the diagonal
    for j in range(3):
        rc, gs = om.fvsc("GaussVolPoint", "grad_s", cell[:, j], bnd[:, j])
        assert np.allclose(gs[tri][:, j], g[:, 0, j], rtol=1e-13, atol=1e-13)


def test_scheme_checks():
    """[fvsc.C:60-63]"""
    mesh = make_mesh("box654"); om = oracle_mesh_of(mesh)
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 1)
    assert om.fvsc("leastSquares", "grad_s", cell, bnd)[0] == -4
    assert om.fvsc("leastSquaresOpt", "grad_s", cell, bnd)[0] == -4
    assert om.fvsc("noSuch", "grad_s", cell, bnd)[0] == -5
    assert om.fvsc("GaussVolPoint", "grad_s", cell, bnd)[0] == 0


def _case(kind, scheme, bc=None, **opt):
    mesh = make_mesh(kind); om = oracle_mesh_of(mesh)
    oc = OracleCase(om, q.default_options(stencil=scheme, **opt))
    if bc:
        bc(oc)
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    oc.set_fields(U, T, p)
    return mesh, om, oc


def test_explicit_branch_keeps_U_consistent_with_rhoU():
    """Property 6 [QGDUEqn.H:79-86]: solve(ddt(rho,U) - ddt(rhoU)) leaves U = rhoU/rho"""
    mesh, om, oc = _case("box654_jitter", "GaussVolPoint", deltaT=1e-3, mu=1e-3)
    oc.step(5)
    U, rho, rhoU = oc.field("U"), oc.field("rho"), oc.field("rhoU")
    assert np.abs(U - rhoU / rho[:, None]).max() < 1e-14


def test_mass_conservation():
    """Property 7 [QGDRhoEqn.H:40-47]: sum V rho changes only through the boundary phiJm"""
    mesh, om, oc = _case("box654", "GaussVolPoint", deltaT=2e-3, mu=1e-3)
    V = om.array("V")
    oc.updateFluxes()
    phiJm = oc.field("phiJm")
    m0 = (oc.field("rho") * V).sum()
    oc.step(1)
    m1 = (oc.field("rho") * V).sum()
    assert abs((m1 - m0) + 2e-3 * phiJm[mesh.nInternalFaces:].sum()) < 1e-15


def test_uniform_state_is_a_fixed_point():
    mesh = make_mesh("box654_jitter"); om = oracle_mesh_of(mesh)
    oc = OracleCase(om, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3))
    n = mesh.nCells
    U = np.tile([0.3, -0.2, 0.1], (n, 1))
    oc.set_fields(U, np.ones(n), np.ones(n))
    r0 = oc.field("rho").copy()
    oc.step(3)
    assert np.abs(oc.field("rho") - r0).max() < 1e-14
    assert np.abs(oc.field("U") - U).max() < 1e-14


def test_energy_equation_follows_the_listing():
    """[QGDEEqn.H:67-72] as written: rho e = rho_old e_old + (rhoE - rhoE_old)"""
    mesh, om, oc = _case("box654", "GaussVolPoint", deltaT=2e-3)
    r0, e0, E0 = oc.field("rho").copy(), oc.field("e").copy(), oc.field("rhoE").copy()
    oc.step(1)
    r1, e1, E1 = oc.field("rho"), oc.field("e"), oc.field("rhoE")
    assert np.abs(r1 * e1 - (r0 * e0 + (E1 - E0))).max() < 1e-14


def test_tau_and_qgd_viscosity_closure():
    """A.6 [constScPrModel1.C:103-115] [QGDThermo.C:91-98]"""
    mesh, om, oc = _case("box654_jitter", "GaussVolPoint", deltaT=1e-3, mu=2e-3, Pr=0.7, ScQGD=0.8, PrQGD=0.9, alphaQGD=0.4)
    c, h, p = oc.field("c"), oc.field("hQGD"), oc.field("p")
    assert np.allclose(oc.field("tauQGD"), 0.4 * h / c, rtol=1e-15)
    assert np.allclose(oc.field("muQGD"), p * 0.8 * oc.field("tauQGD"), rtol=1e-15)
    assert np.allclose(oc.field("mu"), 2e-3 + oc.field("muQGD"), rtol=1e-15)
    assert np.allclose(oc.field("alphau"), 2e-3 / 0.7 + oc.field("muQGD") / 0.9, rtol=1e-14)
    gamma = (q.default_options().Cv + q.default_options().R) / q.default_options().Cv
    assert np.allclose(c, np.sqrt(gamma * q.default_options().R * oc.field("T")), rtol=1e-14)
    oc.updateFluxes()
    own, nei, w = mesh.array("owner"), mesh.array("neighbour"), om.array("weights")
    nif = mesh.nInternalFaces
    aoc = 0.4 / c
    lin = w[:nif] * (aoc[own[:nif]] - aoc[nei]) + aoc[nei]
    assert np.allclose(oc.field("tauQGDf")[:nif], lin * oc.field("hQGDf")[:nif], rtol=1e-15)
    # hQGDf = 2 min(|C_O - C_f|, |C_N - C_f|) [QGDCoeffs.C:303-308]
    C = om.array("C").reshape(-1, 3); Cf = om.array("Cf").reshape(-1, 3)
    ho = np.linalg.norm(C[own[:nif]] - Cf[:nif], axis=1); hn = np.linalg.norm(C[nei] - Cf[:nif], axis=1)
    assert np.allclose(oc.field("hQGDf")[:nif], 2 * np.minimum(ho, hn), rtol=1e-14)


def test_qgdflux_bc_sets_the_wall_mass_flux():
    """[qgdFluxFvPatchScalarField.C:184-192] with GaussVolPoint (quirk B6): on qgdFlux patches with slip walls
    jm.S = rhoU.S - phiwStar - tau S.grad(p); the BC makes the patch-normal pressure gradient cancel phiwStar."""
    mesh = make_mesh("step2d"); om = oracle_mesh_of(mesh)
    oc = OracleCase(om, q.default_options(stencil="GaussVolPoint", deltaT=5e-4))
    cases.forward_step_bcs(oc)
    C = mesh.array("C").reshape(-1, 3)
    U = np.zeros((mesh.nCells, 3)); U[:, 0] = 3.0
    oc.set_fields(U, 1.0 + 0.05 * np.sin(2 * C[:, 0]), 1.0 + 0.05 * np.cos(C[:, 0] + C[:, 1]))
    oc.step(2)
    oc.updateFluxes()
    ps, pz = mesh.array("patchStart"), mesh.array("patchSize")
    phiw, tau, magSf = oc.field("phiwStar"), oc.field("tauQGDf"), om.array("magSf")
    p, pb = oc.field("p"), oc.field("p.boundary")
    own = mesh.array("owner"); dc = om.array("deltaCoeffs")
    for wall in (2, 3, 4):
        f = np.arange(ps[wall], ps[wall] + pz[wall])
        b = f - mesh.nInternalFaces
        grad = dc[f] * (pb[b] - p[own[f]])
        assert np.allclose(grad, -phiw[f] / tau[f] / magSf[f], rtol=1e-9, atol=1e-12)


def wedge_prism_mesh(wedge=True):
    """one prism cell: triangles bottom/top on a generic patch, the three quadrilaterals split between a generic and a
    wedge patch"""
    import qgdsolver_amd as q
    from qgdsolver_amd import _lib as L
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1]], float)
    faces = [[0, 2, 1], [3, 4, 5], [0, 1, 4, 3], [1, 2, 5, 4], [2, 0, 3, 5]]
    return q.PolyMesh.from_arrays(pts, np.cumsum([0] + [len(f) for f in faces]), np.concatenate(faces), np.zeros(5, np.int32),
                                  np.zeros(0, np.int32), 1, [0, 3], [3, 2], [L.PATCH_GENERIC, L.PATCH_WEDGE if wedge else L.PATCH_GENERIC])


def test_gaussvolpoint_refused_on_wedge_meshes_with_prisms():
    """fvsc.C L65-82: GaussVolPoint on a wedge mesh with prism cells is fatal; other stencils and non-wedge meshes are not"""
    from oracle import OracleMesh
    om = OracleMesh(wedge_prism_mesh(True).primitives())
    assert om.fvsc("GaussVolPoint", "grad_s", np.ones(1), np.ones(5))[0] == -4
    assert om.fvsc("reduced", "grad_s", np.ones(1), np.ones(5))[0] == 0
    om2 = OracleMesh(wedge_prism_mesh(False).primitives())
    assert om2.fvsc("GaussVolPoint", "grad_s", np.ones(1), np.ones(5))[0] == 0
