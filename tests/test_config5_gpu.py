"""BASELINE config 5 ("QHDFoam buoyant cavity, 16M unstructured polyhedral cells: irregular stencil stress") on the device.

No polyhedral mesher exists here, so the mesh is the stand-in SURVEY 8(d) C5 names: a 252^3 hexahedral box with the
vertices jittered by 0.2 h (seed 2024), every 7th quad split into two triangles (cells with 6-8 faces), cell labels
shuffled within chunks of 64^3 and then put into the library's Morton order (qgd_mesh_renumber(qgd_mesh_morton_order())).
On it the QHDFoam flux assembly (qgd_qhd_fluxes = QHDFoam/updateFields.H L36-73, updateFluxes.H L33-38, QHDUEqn.H L36-43,
QHDTEqn.H L65-66) runs with the T0byGr / HbyUQHD closures for tauQGDf, and is checked

  * at full size through properties that need no oracle run: phiu = Sf . Uf with the linear weights; a uniform state
    gives zero gradients and phiwo = -tau beta T0 (Sf . g); the triangle pattern of grad(U) (reference quirk B2);
  * at full size against the ORACLE ON A CUT OF THE SAME MESH: a range of ~12 000 cells out of the middle of the 16 M,
    extracted with its vertex-connected ghost layer by qgd_mesh_shard, carries the complete stencil of every face that
    touches an owned cell, so the oracle run on the cut must reproduce the device's full-size result on those faces;
  * against the oracle on a whole 13 824-cell mesh of the same recipe (jitter + triangles + polygon faces + shuffled
    labels + Morton order), all ten outputs, <= 1e-12.
"""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam

import oracle
from util import oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu

BETA, G = 3.4e-3, (0.0, -9.81, 0.0)


_C5_CACHE = {}


def c5_mesh(n, chunk, poly=False):
    """the C5 stand-in recipe at edge n (the 16 M-cell mesh takes 30 s to build: the tests of this module share it)"""
    key = (n, chunk, poly)
    if key not in _C5_CACHE:
        _C5_CACHE.clear()
        _C5_CACHE[key] = _c5_mesh(n, chunk, poly)
    return _C5_CACHE[key]


from qgdsolver_amd.synthetic import c5_mesh as _c5_mesh  # noqa: E402  (the recipe lives in the package: bench.py builds the same mesh)


def cavity_fields(mesh, seed=5):
    """buoyant-cavity-like state: hot xMin wall, cold xMax wall, a smooth velocity + noise so that nothing cancels"""
    rng = np.random.default_rng(seed)
    C = mesh.array("C").reshape(-1, 3)
    Cf = mesh.array("Cf").reshape(-1, 3)[mesh.nInternalFaces:]
    n, nb = mesh.nCells, mesh.nBoundaryFaces

    def vel(X):
        x, y, z = X[:, 0], X[:, 1], X[:, 2]
        return np.stack([np.sin(np.pi * x) * np.cos(np.pi * y), -np.cos(np.pi * x) * np.sin(np.pi * y), 0.1 * np.sin(2 * np.pi * z)], axis=1)

    U = (0.1 * vel(C) + 1e-3 * rng.standard_normal((n, 3)), np.zeros((nb, 3)))           # no-slip walls
    T = (300.0 + 10.0 * (0.5 - C[:, 0]) + 0.1 * rng.standard_normal(n), 300.0 + 10.0 * (0.5 - Cf[:, 0]))
    p = (1e-2 * np.cos(np.pi * C[:, 1]) + 1e-4 * rng.standard_normal(n), 1e-2 * np.cos(np.pi * Cf[:, 1]))
    rho = (np.ones(n), np.ones(nb))
    return U, T, p, rho


def test_c5_recipe_small_mesh_matches_oracle():
    mesh = c5_mesh(24, 8 ** 3, poly=True)
    assert mesh.nCells == 13824
    sizes = np.diff(mesh.array("faceOffsets"))
    assert (sizes == 3).any() and (sizes == 4).any() and (sizes > 4).any()
    om = oracle_mesh_of(mesh)
    dev = q.Device(mesh)
    U, T, p, rho = cavity_fields(mesh)
    phi = np.random.default_rng(2).standard_normal(mesh.nFaces) * 1e-4
    for model, par in (("T0byGr", dict(T0=1.0, Gr=1.0e3)), ("HbyUQHD", dict(aQGD=0.5, UQHD=0.1))):
        tau = qhdfoam.tauQGDf(dev, model, **par)
        got = qhdfoam.updateFluxes(dev, "GaussVolPoint", U, T, rho, tau, BETA, G, p=p, phi=phi)
        ref = oracle.qhd_fluxes(om, "GaussVolPoint", U, T, rho, tau, BETA, G, p=p, phi=phi)
        assert len(ref) == 10
        for k in ref:
            assert rel_err(got[k], ref[k]) <= 1e-12, (model, k, rel_err(got[k], ref[k]))
    dev.close()


def test_c5_full_size_16M_irregular_cells():
    n = 252
    mesh = c5_mesh(n, 64 ** 3)
    assert mesh.nCells == n ** 3
    nif, nf, nc = mesh.nInternalFaces, mesh.nFaces, mesh.nCells
    sizes = np.diff(mesh.array("faceOffsets"))
    tri = sizes == 3
    assert 0.2 < tri.mean() < 0.3 and (sizes[~tri] == 4).all()   # every 7th quad became two triangles
    own, nei, w = mesh.array("owner"), mesh.array("neighbour"), mesh.array("weights")
    Sf = mesh.array("Sf").reshape(-1, 3)
    dev = q.Device(mesh)
    tau = qhdfoam.tauQGDf(dev, "T0byGr", T0=1.0, Gr=1.0e3)
    assert np.array_equal(tau[:nif], np.full(nif, 1e-3))

    # ---- a uniform state: zero gradients, phiwo = -tau beta T0 (Sf . g) --------------------------------------------
    U0 = (np.tile([0.3, -0.2, 0.1], (nc, 1)), np.tile([0.3, -0.2, 0.1], (nf - nif, 1)))
    T0 = (np.full(nc, 300.0), np.full(nf - nif, 300.0))
    one = (np.ones(nc), np.ones(nf - nif))
    got = qhdfoam.updateFluxes(dev, "GaussVolPoint", U0, T0, one, tau, BETA, G)
    h = 1.0 / n
    assert np.abs(got["gradUf"]).max() <= 1e-12 * 0.3 / h
    assert np.abs(got["gradTf"]).max() <= 1e-12 * 300.0 / h
    assert np.allclose(got["phiu"], Sf @ np.array([0.3, -0.2, 0.1]), rtol=1e-13, atol=1e-18)
    expect = -(tau * (BETA * 300.0)) * (Sf @ np.array(G))
    assert np.abs(got["phiwo"] - expect).max() <= 1e-10 * np.abs(expect).max()
    del got, U0, T0, expect

    # ---- the cavity state ------------------------------------------------------------------------------------------
    U, T, p, rho = cavity_fields(mesh)
    phi = np.random.default_rng(2).standard_normal(nf) * 1e-6
    tau = qhdfoam.tauQGDf(dev, "HbyUQHD", aQGD=0.5, UQHD=0.1)
    got = qhdfoam.updateFluxes(dev, "GaussVolPoint", U, T, rho, tau, BETA, G, p=p, phi=phi)
    # phiu = Sf . Uf, Uf by linear interpolation [QHDFoam/updateFluxes.H L33]
    Uf = w[:nif, None] * (U[0][own[:nif]] - U[0][nei]) + U[0][nei]
    ref_phiu = (Sf[:nif] * Uf).sum(1)
    assert np.abs(got["phiu"][:nif] - ref_phiu).max() <= 1e-13 * np.abs(ref_phiu).max()
    assert np.array_equal(got["taubyrhof"][:nif], tau[:nif])
    assert np.abs(got["phiTf"][:nif] - phi[:nif] * (w[:nif] * (T[0][own[:nif]] - T[0][nei]) + T[0][nei])).max() <= 1e-13 * 300e-6
    del Uf, ref_phiu
    # interior triangles: every row of grad(U) holds (dxUx, dyUy, dzUz) [GaussVolPointBase3D.C L844-854]
    gU = got["gradUf"][:nif][tri[:nif]]
    assert np.array_equal(gU[:, 0:3], gU[:, 3:6]) and np.array_equal(gU[:, 0:3], gU[:, 6:9])
    quad_rows = got["gradUf"][:nif][~tri[:nif]][:100000]
    assert not np.array_equal(quad_rows[:, 0:3], quad_rows[:, 3:6])
    del gU, quad_rows

    # ---- the oracle on a cut of this very mesh ---------------------------------------------------------------------
    a = nc // 2 + 12345
    cut = mesh.shard(3, 1, cell_start=[0, a, a + 12000, nc])
    cg, fg = cut.array("cellGlobal"), cut.array("faceGlobal")
    assert cut.nCells < 60000
    flipped = fg < 0
    gface = np.where(flipped, -1 - fg, fg)
    cnif = cut.nInternalFaces
    gb = gface[cnif:] - nif              # boundary index in the full mesh; negative: a cut (halo) face
    real = gb >= 0
    assert not flipped[cnif:][real].any()

    def restrict(pair):
        cell, bnd = pair
        b = np.zeros((cut.nBoundaryFaces,) + bnd.shape[1:])
        b[real] = bnd[gb[real]]
        return cell[cg], b

    om = oracle_mesh_of(cut)
    ref = oracle.qhd_fluxes(om, "GaussVolPoint", restrict(U), restrict(T), restrict(rho), tau[gface], BETA, G, p=restrict(p),
                            phi=np.where(flipped, -phi[gface], phi[gface]))
    cown, cnei = cut.array("owner"), cut.array("neighbour")
    owned = (cg >= a) & (cg < a + 12000)
    sel = np.zeros(cut.nFaces, bool)
    sel[:cnif] = owned[cown[:cnif]] | owned[cnei]           # complete stencil: the face touches an owned cell
    sel[cnif:] = real & owned[cown[cnif:]]
    assert sel.sum() > 3 * 12000
    odd = {"phiu", "phiwo", "phiUf", "phiTf", "phiTauTReg"}   # fluxes change sign with the face orientation
    for k in ref:
        g = got[k][gface]
        if k in odd:
            g = np.where(flipped.reshape((-1,) + (1,) * (g.ndim - 1)), -g, g)
        scale = np.abs(ref[k][sel]).max()
        err = np.abs(g[sel] - ref[k][sel]).max() / scale
        assert err <= 1e-11, (k, err)
    del got, ref

    # ---- config 5 end to end: the resident QHDFoam step on this mesh (buoyant cavity, pressure equation included) ------
    from test_qhd_case import cavity_bcs, divergence
    case = qhdfoam.QHDFoamCase(dev, qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=0.1, rho0=1.0, mu=1e-3,
                                                        Pr=0.71, beta=BETA, g=G, deltaT=0.02 * h / 0.1, pTol=1e-8, pMaxIter=300, pRefCell=0))
    cavity_bcs(case, mesh)
    case.set_fields(U[0], T[0], p[0])
    case.step(3)
    info = case.info()
    print(f"C5 QHDFoam step, {nc} cells: {info}")
    assert info["steps"] == 3 and info["pFinalResidual"] < 1e-8 and 0 < info["pIterations"] < 150, info
    phi3 = case.field("phi")
    assert np.abs(divergence(mesh, phi3)).max() <= 1e-5 * np.abs(phi3).max()
    assert np.abs(phi3[nif:]).max() <= 1e-12 * np.abs(phi3).max()          # impermeable walls
    Un, Tn = case.field("U"), case.field("T")
    assert np.isfinite(Un).all() and np.isfinite(Tn).all() and np.abs(Un).max() < 1.0 and 289.0 < Tn.min() and Tn.max() < 311.0
    case.close()
    dev.close()


def test_c5_qhd_case_cut_8_ways_resident_on_one_gpu():
    """Config 5 as configured for 8 GPUs, on one: the 16 M-cell mesh cut into 8 cell ranges (qgd_mesh_shard: every range with its
    vertex-connected ghost layer and one halo slot per neighbouring range), eight resident QHDFoam cases stepped in lockstep by the
    QhdStepper a real run uses (messages device-to-device through the pack / unpack kernels, the PCG's sums through the control
    blocks), two steps, against the unsharded 16 M-cell case.  The pressure equation is solved to 1e-11, so p agrees to ~1e-7 and
    U, T to 1e-9.  The pressure solver's multigrid hierarchy spans the eight shards (two all-reduces gather the global matrix at the first
    step; level 0 distributed, coarse levels replicated), so the iteration count is the unsharded one."""
    import os
    from qgdsolver_amd.halo import LocalWorld, QhdStepper
    from qhd_shards import gather, range_shards
    from test_qhd_case import cavity_bcs, options
    n = int(os.environ.get("QGD_C5_EDGE", "252"))
    world, steps = 8, 2
    mesh = c5_mesh(n, 64 ** 3)
    nc = mesh.nCells
    C = mesh.array("C").reshape(-1, 3)
    rng = np.random.default_rng(11)
    fields = (1e-3 * rng.standard_normal((nc, 3)), 300.0 + 10.0 * (0.5 - C[:, 0]), np.zeros(nc))
    opt = options("GaussVolPoint", deltaT=0.2 / n, pTol=1e-11, pMaxIter=600, pRefCell=nc // 3, pRefValue=0.0)
    dev = q.Device(mesh)
    whole = qhdfoam.QHDFoamCase(dev, opt)
    cavity_bcs(whole, mesh)
    whole.set_fields(*fields)
    whole.step(steps)
    want = {f: whole.field(f) for f in ("U", "T", "p")}
    winfo = whole.info()
    whole.close(); dev.close()
    assert winfo["pFinalResidual"] < 1e-11 and 0 < winfo["pIterations"] < 100, winfo
    shards = range_shards(mesh, world)
    assert sum(len(sh["owned"]) for sh in shards) == nc
    assert all(len([p for p in sh["peers"] if p >= 0]) >= 1 for sh in shards)
    pairs = []
    for sh in shards:
        d = q.Device(sh["mesh"])
        c = qhdfoam.QHDFoamCase(d, opt)
        cavity_bcs(c, sh["mesh"])
        cg = sh["cell_global"]
        c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg])
        pairs.append((d, c))
    cases = [c for _, c in pairs]
    QhdStepper(LocalWorld(cases, [sh["peers"] for sh in shards])).step(steps)
    infos = [c.info() for c in cases]
    assert all(i["steps"] == steps and i["pFinalResidual"] < 1e-11 for i in infos), infos
    # the multigrid hierarchy spans the ranks (distributed level 0, replicated coarse levels): the sharded solve needs the iterations
    # of the unsharded one (16 = 16 at 1e-11; with rank-local hierarchies, QGD_MG_DIST=0, it was 186)
    assert len({i["pIterations"] for i in infos}) == 1 and infos[0]["pIterations"] <= winfo["pIterations"] + 2, (infos[0], winfo)
    print("c5 8-way: pressure iterations", infos[0]["pIterations"], "sharded,", winfo["pIterations"], "unsharded")
    for f, ncomp, tol in (("U", 3, 1e-8), ("T", 1, 1e-9), ("p", 1, 1e-6)):
        got = gather(shards, cases, f, nc, ncomp)
        err = np.abs(got - want[f]).max() / np.abs(want[f]).max()
        assert err <= tol, (f, err, infos[0], winfo)
    for d, c in pairs:
        c.close(); d.close()
