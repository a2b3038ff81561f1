"""Cell-range shards of arbitrary meshes on the device: W shards share one GPU in this process, halo messages go
through device buffers (pack on the sender's case, unpack on the receiver's), and the owned cells reproduce the
unsharded device run and the oracle."""
import numpy as np
import pytest

import qgdsolver_amd as q
import cases
from test_partition import mixed_bcs, random_perm, rcm_poly_mesh, run_oracle
from util import make_mesh

pytestmark = pytest.mark.gpu


def run_device(mesh, stencil, bc_fn, U, T, p, steps, fused_tables=True, **opt):
    dev = q.Device(mesh, fused_tables=fused_tables)
    gc = q.QGDFoamCase(dev, q.default_options(stencil=stencil, **opt))
    if fused_tables in ("any", False):
        assert gc.fused_info()["fused"] == (fused_tables == "any"), gc.fused_info()
    if bc_fn:
        bc_fn(gc)
    gc.set_fields(U, T, p)
    gc.step(steps)
    out = {f: gc.field(f) for f in ("rho", "U", "p", "e")}
    gc.close(); dev.close()
    return out


def run_sharded_device(g, world, stencil, bc_fn, U, T, p, steps, overlapped=False, fused_tables=True, **opt):
    """fused_tables="any": every shard steps with the fused one-launch kernel (phase 1 = all blocks; phases 10 / 11 = the blocks that hold a
    cell a neighbour waits for, then the others, with the record buffers swapped in between) -- asserted per shard; False: the separate
    kernels; True: the library's choice (shards this small take the separate kernels)"""
    shards = [g.shard(world, r) for r in range(world)]
    devs, cs = [], []
    for s in shards:
        d = q.Device(s, fused_tables=fused_tables)
        c = q.QGDFoamCase(d, q.default_options(stencil=stencil, **opt))
        if fused_tables in ("any", False):
            fi = c.fused_info()
            assert fi["fused"] == (fused_tables == "any"), fi
            assert fused_tables is False or 1 <= fi["layerBlocks"] <= fi["blocks"], fi
        if bc_fn:
            bc_fn(c)
        cg = s.array("cellGlobal")
        c.set_fields(U[cg], T[cg], p[cg])
        devs.append(d); cs.append(c)
    bufs = {}
    for r, (s, c, d) in enumerate(zip(shards, cs, devs)):
        for k, peer in enumerate(s.array("haloPeer")):
            bufs[(r, int(peer))] = d.alloc(8 * max(1, c.halo_count(k)))
    with pytest.raises(q.QgdError):
        cs[0].step(1)  # a sharded case is driven phase by phase

    def exchange():
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                c.halo_pack(k, bufs[(r, int(peer))])
            c.sync()
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                assert c.halo_recv_count(k) == cs[int(peer)].halo_count(list(shards[int(peer)].array("haloPeer")).index(r))
                c.halo_unpack(k, bufs[(int(peer), r)])
            c.step_phase(2)
            c.sync()

    mid = {}
    if cs[0].needs_mid_exchange():
        for r, (s, c, d) in enumerate(zip(shards, cs, devs)):
            for k, peer in enumerate(s.array("haloPeer")):
                mid[(r, int(peer))] = d.alloc(8 * max(1, c.mid_halo_count(k)[0]))

    def exchange_mid():
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                c.mid_halo_pack(k, mid[(r, int(peer))])
            c.sync()
        for r, (s, c) in enumerate(zip(shards, cs)):
            for k, peer in enumerate(s.array("haloPeer")):
                c.mid_halo_unpack(k, mid[(int(peer), r)])
            c.sync()

    exchange()
    for _ in range(steps):
        if mid:
            for c in cs:
                c.step_phase(5)
            exchange_mid()
            for c in cs:
                c.step_phase(6)
        else:
            for c in cs:
                c.step_phase(0)
        if overlapped:  # boundary layer first, exchange, then the rest: same result as the plain order
            for c in cs:
                c.step_phase(10)
            exchange()
            for c in cs:
                c.step_phase(11)
                c.sync()
        else:
            for c in cs:
                c.step_phase(1)
            exchange()
    out = {f: np.zeros_like(U if f == "U" else T, dtype=float) for f in ("rho", "U", "p", "e")}
    for r, (s, c) in enumerate(zip(shards, cs)):
        cg = s.array("cellGlobal")
        own = (cg >= (g.nCells * r) // world) & (cg < (g.nCells * (r + 1)) // world)
        for f in out:
            out[f][cg[own]] = c.field(f)[own]
    for c, d in zip(cs, devs):
        c.close(); d.close()
    return out


@pytest.mark.parametrize("kind,stencil,world,bc_fn,opt,overlapped,fused", [
    ("box654_poly", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), False, False),
    ("box654_poly", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), True, False),
    ("box654_poly_rcm", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), False, False),
    ("box654_poly_rcm", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), True, False),
    # the fused one-launch step on shards against the ORACLE (not only against the separate kernels, tests/test_fused_step_gpu.py): plain
    # order (phases 0, 1) and boundary-layer-first order (0, 10, 11), 2 and 3 shards, meshes with triangles / polygons / qgdFlux walls
    ("box654_poly", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), False, "any"),
    ("box654_poly", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), True, "any"),
    ("box654_poly_rcm", "GaussVolPoint", 2, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), True, "any"),
    ("box12108", "GaussVolPoint", 2, None, dict(deltaT=2e-3, mu=1e-3), True, "any"),
    ("box12108", "GaussVolPoint", 3, mixed_bcs, dict(deltaT=1e-3, mu=1e-3), False, "any"),
    ("box654_jitter", "reduced", 2, mixed_bcs, dict(deltaT=1e-3), False, True),
    ("step2d", "leastSquares", 4, cases.forward_step_bcs, dict(deltaT=5e-4), True, True),
    ("step2d", "GaussVolPoint", 3, cases.forward_step_bcs, dict(deltaT=5e-4), False, True),
])
def test_sharded_device_matches_unsharded_and_oracle(kind, stencil, world, bc_fn, opt, overlapped, fused):
    g = rcm_poly_mesh() if kind == "box654_poly_rcm" else (q.PolyMesh.box(12, 10, 8).jitter(0.1, seed=5) if kind == "box12108" else make_mesh(kind))
    if kind == "box654_poly":
        g.renumber(random_perm(g.nCells, 21))
    C = g.array("C").reshape(-1, 3)
    if kind == "step2d":
        U = np.zeros((g.nCells, 3)); U[:, 0] = 3.0
        T = 1.0 + 0.05 * np.sin(2.0 * C[:, 0]) * np.cos(3.0 * C[:, 1])
        p = 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])
    else:
        U, T, p = cases.box_initial_fields(C)
    steps = 10
    ref = run_oracle(g, stencil, bc_fn, U, T, p, steps, **opt)
    one = run_device(g, stencil, bc_fn, U, T, p, steps, fused_tables=fused, **opt)
    got = run_sharded_device(g, world, stencil, bc_fn, U, T, p, steps, overlapped=overlapped, fused_tables=fused, **opt)
    for f in ref:
        scale = np.abs(ref[f]).max()
        assert np.abs(got[f] - one[f]).max() <= 1e-12 * scale, (kind, stencil, f, "sharded vs unsharded device")
        assert np.abs(got[f] - ref[f]).max() <= 1e-10 * scale, (kind, stencil, f, "sharded device vs oracle")


def test_renumbered_mesh_on_the_device():
    """RCM-renumbered and randomly renumbered meshes give the same cells the same state on the device"""
    g = make_mesh("box654_tri")
    U, T, p = cases.box_initial_fields(g.array("C").reshape(-1, 3))
    a = run_device(g, "GaussVolPoint", mixed_bcs, U, T, p, 10, deltaT=1e-3, mu=1e-3)
    perm = random_perm(g.nCells, 4)
    g.renumber(perm)
    order = g.rcm_order()
    g.renumber(order)
    total = order[perm]  # old -> new after both
    inv = np.argsort(total)
    b = run_device(g, "GaussVolPoint", mixed_bcs, U[inv], T[inv], p[inv], 10, deltaT=1e-3, mu=1e-3)
    for f in a:
        assert np.abs(b[f][total] - a[f]).max() <= 1e-12 * np.abs(a[f]).max(), f
