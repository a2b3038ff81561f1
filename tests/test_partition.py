"""Cell renumbering, reverse Cuthill-McKee and cell-range sharding of arbitrary meshes: invariants of
the mesh operations, and the sharded ORACLE (halo messages moved by hand, in process) against the unsharded one."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
import cases
from oracle import OracleCase, OracleMesh
from util import make_mesh, oracle_mesh_of


def random_perm(n, seed):
    return np.random.default_rng(seed).permutation(n).astype(np.int32)


@pytest.mark.parametrize("kind", ["box654_jitter", "box654_poly", "step2d"])
def test_renumber_keeps_the_mesh(kind):
    mesh = make_mesh(kind)
    ref = {k: mesh.array(k).copy() for k in ("V", "C", "Sf", "Cf", "owner", "neighbour", "patchStart", "patchSize")}
    nIF = mesh.nInternalFaces
    perm = random_perm(mesh.nCells, 3)
    fmap = mesh.renumber(perm)
    own, nei = mesh.array("owner"), mesh.array("neighbour")
    assert np.all(nei > own[:nIF])
    key = own[:nIF].astype(np.int64) * mesh.nCells + nei
    assert np.all(np.diff(key) >= 0), "internal faces must be in upper-triangular order"
    assert np.array_equal(mesh.array("patchStart"), ref["patchStart"]) and np.array_equal(mesh.array("patchSize"), ref["patchSize"])
    V, Cc = mesh.array("V"), mesh.array("C").reshape(-1, 3)
    assert np.allclose(V[perm], ref["V"], rtol=1e-13, atol=0) and np.allclose(Cc[perm], ref["C"].reshape(-1, 3), rtol=1e-12, atol=1e-15)
    Sf, Cf = mesh.array("Sf").reshape(-1, 3), mesh.array("Cf").reshape(-1, 3)
    new = np.where(fmap >= 0, fmap, -1 - fmap)
    sign = np.where(fmap >= 0, 1.0, -1.0)[:, None]
    assert sorted(new) == list(range(mesh.nFaces))
    assert np.allclose(Sf[new], sign * ref["Sf"].reshape(-1, 3), rtol=1e-12, atol=1e-16)
    assert np.allclose(Cf[new], ref["Cf"].reshape(-1, 3), rtol=1e-12, atol=1e-16)
    # cells follow their faces
    assert np.array_equal(np.where(fmap[:nIF] >= 0, own[new[:nIF]], nei[new[:nIF]]), perm[ref["owner"][:nIF]])
    with pytest.raises(q.QgdError):
        mesh.renumber(np.zeros(mesh.nCells, dtype=np.int32))


def run_oracle(mesh, stencil, bc_fn, U, T, p, steps, **opt):
    oc = OracleCase(oracle_mesh_of(mesh), q.default_options(stencil=stencil, **opt))
    if bc_fn:
        bc_fn(oc)
    oc.set_fields(U, T, p)
    oc.step(steps)
    return {f: oc.field(f) for f in ("rho", "U", "p", "e")}


def mixed_bcs(case):
    case.set_bc(0, U=("fixedValue", (0.1, 0.0, 0.0)), T=("fixedValue", 1.05), p=("zeroGradient", None))
    case.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
    case.set_bc(2, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))
    case.set_bc(3, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))


def test_solution_does_not_depend_on_the_cell_numbering():
    """the same case on a randomly renumbered mesh gives the same cells the same state (summation order aside)"""
    mesh = make_mesh("box654_poly")
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    a = run_oracle(mesh, "GaussVolPoint", mixed_bcs, U, T, p, 8, deltaT=1e-3, mu=1e-3)
    perm = random_perm(mesh.nCells, 11)
    mesh.renumber(perm)
    inv = np.argsort(perm)  # old label of each new cell
    b = run_oracle(mesh, "GaussVolPoint", mixed_bcs, U[inv], T[inv], p[inv], 8, deltaT=1e-3, mu=1e-3)
    for f in a:
        assert np.abs(b[f][perm] - a[f]).max() <= 1e-12 * np.abs(a[f]).max(), f


def bandwidth(mesh):
    return int(np.abs(mesh.array("neighbour").astype(np.int64) - mesh.array("owner")[:mesh.nInternalFaces]).max())


def test_rcm_reduces_the_bandwidth_of_a_shuffled_mesh():
    mesh = q.PolyMesh.box(12, 10, 8)
    natural = bandwidth(mesh)
    mesh.renumber(random_perm(mesh.nCells, 5))
    shuffled = bandwidth(mesh)
    order = mesh.rcm_order()
    assert sorted(order) == list(range(mesh.nCells))
    mesh.renumber(order)
    assert bandwidth(mesh) < shuffled / 3 and bandwidth(mesh) <= 2 * natural


def test_morton_order_restores_locality():
    """Z-curve order of the cell centres: a permutation that brings face neighbours (and, through renumber, the
    vertices of a cell) back together after a random relabelling"""
    mesh = q.PolyMesh.box(16, 16, 16)
    mesh.renumber(random_perm(mesh.nCells, 8))

    def mean_gap(m):
        return float(np.abs(m.array("neighbour").astype(np.int64) - m.array("owner")[:m.nInternalFaces]).mean())

    def vertex_spread(m):
        fp, fo = m.array("facePoints").astype(np.int64), m.array("faceOffsets")
        quads = fp.reshape(-1, 4)  # a hex box has quads only
        assert fo[-1] == quads.size
        return float((quads.max(axis=1) - quads.min(axis=1)).mean())

    gap0, spread0 = mean_gap(mesh), vertex_spread(mesh)
    order = mesh.morton_order()
    assert sorted(order) == list(range(mesh.nCells))
    mesh.renumber(order)
    assert mean_gap(mesh) < gap0 / 5 and vertex_spread(mesh) < spread0 / 5
    # 8 consecutive labels form a 2x2x2 brick
    Cc = mesh.array("C").reshape(-1, 3)
    brick = Cc[:8]
    assert np.allclose(brick.max(axis=0) - brick.min(axis=0), 1.0 / 16, atol=1e-12)


def shards_of(gmesh, world, cell_start=None):
    return [gmesh.shard(world, r, cell_start) for r in range(world)]


@pytest.mark.parametrize("kind,world,shuffle", [("box654_poly", 3, True), ("step2d", 4, False), ("box654_jitter", 2, False)])
def test_shard_lists_are_consistent(kind, world, shuffle):
    g = make_mesh(kind)
    if shuffle:
        g.renumber(random_perm(g.nCells, 9))
    shards = shards_of(g, world)
    gV, gC, gSf = g.array("V"), g.array("C").reshape(-1, 3), g.array("Sf").reshape(-1, 3)
    owned_total = 0
    for r, s in enumerate(shards):
        cg = s.array("cellGlobal")
        lo, hi = (g.nCells * r) // world, (g.nCells * (r + 1)) // world
        is_owned = (cg >= lo) & (cg < hi)
        owned_total += int(is_owned.sum())
        assert int(is_owned.sum()) == hi - lo and np.all(np.diff(cg) > 0)
        assert np.allclose(s.array("V"), gV[cg], rtol=1e-13) and np.allclose(s.array("C").reshape(-1, 3), gC[cg], rtol=1e-12, atol=1e-15)
        fg = s.array("faceGlobal")
        lab = np.where(fg >= 0, fg, -1 - fg)
        assert np.allclose(s.array("Sf").reshape(-1, 3), np.where(fg >= 0, 1.0, -1.0)[:, None] * gSf[lab], rtol=1e-12, atol=1e-16)
        assert s.array("patchType")[-1] == L.PATCH_HALO and s.nPatches == g.nPatches + 1
        peers = s.array("haloPeer")
        assert s.halo_slots == len(peers) and r not in peers
        ghosts = np.concatenate([s.array(f"haloGhost{k}") for k in range(len(peers))]) if len(peers) else np.zeros(0, int)
        assert sorted(ghosts) == list(np.nonzero(~is_owned)[0]), "every ghost cell is refreshed by exactly one neighbour"
        for k, peer in enumerate(peers):
            other = shards[peer]
            back = list(other.array("haloPeer")).index(r)
            mine = cg[s.array(f"haloSend{k}")]
            theirs = other.array("cellGlobal")[other.array(f"haloGhost{back}")]
            assert np.array_equal(mine, theirs), (r, peer)
            assert np.all((mine >= lo) & (mine < hi))
    assert owned_total == g.nCells


def run_sharded_oracle(g, world, stencil, bc_fn, U, T, p, steps, cell_start=None, **opt):
    shards = shards_of(g, world, cell_start)
    cs = []
    for s in shards:
        om = OracleMesh(s.primitives())
        for k in range(s.halo_slots):
            om.set_halo(k, s.array(f"haloGhost{k}"), s.array(f"haloSend{k}"))
        c = OracleCase(om, q.default_options(stencil=stencil, **opt))
        if bc_fn:
            bc_fn(c)
        cg = s.array("cellGlobal")
        c.set_fields(U[cg], T[cg], p[cg])
        cs.append(c)

    def exchange():
        bufs = {}
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                buf = np.zeros(c.halo_count(k))
                c.halo_pack(k, buf)
                bufs[(r, int(peer))] = buf
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                buf = bufs[(int(peer), r)]
                assert buf.size == c.halo_recv_count(k)
                c.halo_unpack(k, buf)
            c.step_phase(2)

    def exchange_mid():
        """the message in the middle of the assembly (GaussVolPoint shards that meet a qgdFlux wall, include/qgd_amd.h)"""
        bufs = {}
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                buf = np.zeros(max(c.mid_halo_count(k)[0], 1))
                c.mid_halo_pack(k, buf)
                bufs[(r, int(peer))] = buf
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                assert c.mid_halo_count(k)[1] == cs[int(peer)].mid_halo_count(list(shards[int(peer)].array("haloPeer")).index(r))[0]
                c.mid_halo_unpack(k, bufs[(int(peer), r)])

    exchange()
    for _ in range(steps):
        if cs[0].needs_mid_exchange():
            for c in cs:
                c.step_phase(5)
            exchange_mid()
            for c in cs:
                c.step_phase(6)
        else:
            for c in cs:
                c.step_phase(0)
        if opt.get("adjustTimeStep"):
            red = np.maximum.reduce([c.reduction() for c in cs])
            for c in cs:
                c.reduction(red)
        for c in cs:
            c.step_phase(1)
        exchange()
    out = {f: np.zeros_like(np.atleast_1d(U if f == "U" else T), dtype=float) for f in ("rho", "U", "p", "e")}
    for r, (s, c) in enumerate(zip(shards, cs)):
        cg = s.array("cellGlobal")
        lo = (g.nCells * r) // world if cell_start is None else cell_start[r]
        hi = (g.nCells * (r + 1)) // world if cell_start is None else cell_start[r + 1]
        own = (cg >= lo) & (cg < hi)
        for f in out:
            out[f][cg[own]] = c.field(f)[own]
    return out


def rcm_poly_mesh():
    """box654_poly relabelled at random, then put into reverse Cuthill-McKee order: its 3-way range cut meets the qgdFlux walls where the
    patch faces of ghost cells have incomplete stencils -- without the mid-assembly message the shards are off by 6e-7 in U after 5 steps"""
    g = make_mesh("box654_poly")
    g.renumber(np.random.default_rng(4).permutation(g.nCells).astype(np.int32))
    g.renumber(g.rcm_order())
    return g


@pytest.mark.parametrize("kind,stencil,world,bc_fn,opt", [
    ("box654_poly", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3)),
    ("box654_poly_rcm", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3)),
    ("box654_poly_rcm", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3, adjustTimeStep=1, maxCo=0.3, maxDeltaT=1.0, cTau=0.75)),
    ("box654_jitter", "reduced", 2, mixed_bcs, dict(deltaT=1e-3)),
    ("step2d", "leastSquares", 4, cases.forward_step_bcs, dict(deltaT=5e-4)),
    ("step2d", "GaussVolPoint", 3, cases.forward_step_bcs, dict(deltaT=5e-4, adjustTimeStep=1, maxCo=0.3, maxDeltaT=1.0, cTau=0.75)),
])
def test_sharded_oracle_on_arbitrary_meshes(kind, stencil, world, bc_fn, opt):
    g = rcm_poly_mesh() if kind == "box654_poly_rcm" else make_mesh(kind)
    if kind == "box654_poly":
        g.renumber(random_perm(g.nCells, 21))  # several neighbours per rank, corner cells needed by two of them
    C = g.array("C").reshape(-1, 3)
    if kind == "step2d":
        U = np.zeros((g.nCells, 3)); U[:, 0] = 3.0
        T = 1.0 + 0.05 * np.sin(2.0 * C[:, 0]) * np.cos(3.0 * C[:, 1])
        p = 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])
    else:
        U, T, p = cases.box_initial_fields(C)
    ref = run_oracle(g, stencil, bc_fn, U, T, p, 8, **opt)
    got = run_sharded_oracle(g, world, stencil, bc_fn, U, T, p, 8, **opt)
    for f in ref:
        assert np.abs(got[f] - ref[f]).max() <= 1e-12 * np.abs(ref[f]).max(), (kind, stencil, f)


def case_setup(kind):
    """(global mesh, BC function, initial fields, options) shared with tests/range_halo_worker.py"""
    g = make_mesh(kind)
    if kind == "box654_poly":
        g.renumber(random_perm(g.nCells, 21))
    C = g.array("C").reshape(-1, 3)
    if kind == "step2d":
        U = np.zeros((g.nCells, 3)); U[:, 0] = 3.0
        T = 1.0 + 0.05 * np.sin(2.0 * C[:, 0]) * np.cos(3.0 * C[:, 1])
        p = 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])
        return g, cases.forward_step_bcs, (U, T, p), dict(deltaT=5e-4)
    return g, mixed_bcs, cases.box_initial_fields(C), dict(deltaT=1e-3, mu=1e-3)


def test_uneven_ranges_and_bad_arguments():
    g = make_mesh("box654")
    U, T, p = cases.box_initial_fields(g.array("C").reshape(-1, 3))
    ref = run_oracle(g, "GaussVolPoint", None, U, T, p, 5, deltaT=1e-3)
    got = run_sharded_oracle(g, 3, "GaussVolPoint", None, U, T, p, 5, cell_start=[0, 17, 95, g.nCells], deltaT=1e-3)
    for f in ref:
        assert np.abs(got[f] - ref[f]).max() <= 1e-12 * np.abs(ref[f]).max(), f
    with pytest.raises(q.QgdError):
        g.shard(2, 0, [0, 0, g.nCells])
    with pytest.raises(q.QgdError):
        g.shard(2, 2)
    with pytest.raises(q.QgdError):
        g.shard(2, 0).shard(2, 0)
