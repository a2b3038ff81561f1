"""QGDFoam with implicitDiffusion true -- the reference's default [QGDThermo.C L70-82] -- on cell-range shards:
the advance as phases 20..35 with reductions of the solves' control block and the branch's own halo messages between them
(include/qgd_amd.h "the implicitDiffusion branch on a cell-range shard", qgdsolver_amd.halo.ImplicitStepper).

CPU: the oracle's phases on an UNSHARDED mesh reproduce its monolithic step; 2-3 oracle shards in one process (box slabs, cell
ranges of a renumbered polygonal mesh) against the unsharded oracle; the same over gloo, one rank per shard.
GPU: the device's phases against its own step, HIP shards on one device against the unsharded HIP run and the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd.halo import ImplicitShard, ImplicitStepper, LocalWorld

import cases
from oracle import OracleCase
from qhd_shards import box_slabs, gather, oracle_shard_mesh, range_shards
from test_partition import mixed_bcs
from util import make_mesh, oracle_mesh_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPT = dict(stencil="GaussVolPoint", deltaT=5e-4, mu=2e-2, implicitDiffusion=1, implicitTol=1e-14, implicitMaxIter=500)
FIELDS = (("rho", 1), ("U", 3), ("p", 1), ("e", 1))
KINDS = range(5)


def run_unsharded_oracle(mesh, bc_fn, fields, steps, phases=False):
    oc = OracleCase(oracle_mesh_of(mesh), q.default_options(**OPT))
    if bc_fn:
        bc_fn(oc)
    oc.set_fields(*fields)
    if phases:
        ImplicitStepper(LocalWorld([ImplicitShard(oc)], [[]], kinds=KINDS)).step(steps)
    else:
        oc.step(steps)
    return oc


def oracle_shard_case(sh, bc_fn, fields):
    c = OracleCase(oracle_shard_mesh(sh["mesh"]), q.default_options(**OPT))
    if bc_fn:
        bc_fn(c)
    cg = sh["cell_global"]
    c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg])
    return c


@pytest.mark.parametrize("kind", ["box654_jitter", "plane2d_jitter"])
def test_oracle_phases_are_the_oracle_step(kind):
    mesh = make_mesh(kind)
    fields = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    if mesh.nGeometricD == 2:
        fields[0][:, 2] = 0.0
    bc = mixed_bcs if kind.startswith("box") else None
    a = run_unsharded_oracle(mesh, bc, fields, 5)
    b = run_unsharded_oracle(mesh, bc, fields, 5, phases=True)
    for f, _ in FIELDS:
        assert np.abs(a.field(f) - b.field(f)).max() <= 1e-12 * np.abs(a.field(f)).max(), f
    assert a.info()["steps"] == b.info()["steps"] == 5


CUTS = [("slabs", 2), ("slabs", 3), ("ranges", 3)]


def bcs_of(cut):
    """qgdFlux walls on every cut: the range cut of the RCM-ordered mesh needs the mid-assembly message (include/qgd_amd.h, phases 5 | 6)"""
    return mixed_bcs


def cut_mesh(cut, world):
    if cut == "slabs":
        return q.PolyMesh.box(6, 5, 12), box_slabs(6, 5, 12, world)
    g = make_mesh("box654_poly")
    g.renumber(np.random.default_rng(4).permutation(g.nCells).astype(np.int32))
    g.renumber(g.rcm_order())
    return g, range_shards(g, world)


@pytest.mark.parametrize("cut,world", CUTS)
def test_sharded_oracle_matches_unsharded(cut, world):
    g, shards = cut_mesh(cut, world)
    fields = cases.box_initial_fields(g.array("C").reshape(-1, 3))
    steps = 5
    ref = run_unsharded_oracle(g, bcs_of(cut), fields, steps)
    ocs = [oracle_shard_case(sh, bcs_of(cut), fields) for sh in shards]
    ImplicitStepper(LocalWorld([ImplicitShard(c) for c in ocs], [sh["peers"] for sh in shards], kinds=KINDS)).step(steps)
    for f, nc in FIELDS:
        got, want = gather(shards, ocs, f, g.nCells, nc), ref.field(f)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max(), (f, np.abs(got - want).max())


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_oracle_over_gloo(tmp_path, world):
    steps = 4
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29580 + world), os.path.join(ROOT, "tests", "implicit_halo_worker.py"), str(tmp_path), str(steps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g, _ = cut_mesh("ranges", world)
    ref = run_unsharded_oracle(g, mixed_bcs, cases.box_initial_fields(g.array("C").reshape(-1, 3)), steps)
    covered = 0
    for rank in range(world):
        d = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        covered += d["cells"].size
        for f, _ in FIELDS:
            want = ref.field(f)[d["cells"]]
            assert np.abs(d[f] - want).max() <= 1e-10 * np.abs(ref.field(f)).max(), (rank, f)
    assert covered == g.nCells


# ---- GPU -----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_phases_are_the_device_step():
    mesh = make_mesh("box654_jitter")
    fields = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    dev = q.Device(mesh)
    a, b = q.QGDFoamCase(dev, q.default_options(**OPT)), q.QGDFoamCase(dev, q.default_options(**OPT))
    for c in (a, b):
        mixed_bcs(c)
        c.set_fields(*fields)
    a.step(5)
    ImplicitStepper(LocalWorld([ImplicitShard(b)], [[]], kinds=KINDS)).step(5)
    b.sync()
    for f, _ in FIELDS:
        assert np.array_equal(a.field(f), b.field(f)), f
    ia, ib = a.implicit_info(), b.implicit_info()
    assert ia["solves"] == ib["solves"] and ia["unconverged_steps"] == ib["unconverged_steps"] == 0
    a.close(); b.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cut,world", CUTS + [("slabs-big", 3)])
def test_device_shards_match_unsharded_device_and_oracle(cut, world):
    if cut == "slabs-big":
        g, shards = q.PolyMesh.box(10, 9, 12), box_slabs(10, 9, 12, world)
    else:
        g, shards = cut_mesh(cut, world)
    fields = cases.box_initial_fields(g.array("C").reshape(-1, 3))
    steps = 6
    bc = bcs_of(cut)
    ref = run_unsharded_oracle(g, bc, fields, steps)
    gdev = q.Device(g)
    whole = q.QGDFoamCase(gdev, q.default_options(**OPT))
    bc(whole)
    whole.set_fields(*fields)
    whole.step(steps)
    pairs = []
    for sh in shards:
        dev = q.Device(sh["mesh"])
        c = q.QGDFoamCase(dev, q.default_options(**OPT))
        bc(c)
        cg = sh["cell_global"]
        c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg])
        with pytest.raises(q.QgdError):
            c.step_phase(1)      # on a shard the branch advances through phases 20..35
        pairs.append((dev, c))
    cs = [c for _, c in pairs]
    ImplicitStepper(LocalWorld([ImplicitShard(c) for c in cs], [sh["peers"] for sh in shards], kinds=KINDS)).step(steps)
    for f, nc in FIELDS:
        got = gather(shards, cs, f, g.nCells, nc)
        for tag, want in (("unsharded device", whole.field(f)), ("oracle", ref.field(f))):
            err = np.abs(got - want).max() / np.abs(want).max()
            assert err <= 1e-10, (cut, f, tag, err)
    for c in cs:
        ii = c.implicit_info()
        # the Chebyshev solves measure the TRUE residual b - A x, which stalls at the rounding floor of the product (~1e-14 here for e, whose
        # normFactor is small against |b|); the solver stops there instead of burning maxIter steps, and says so
        assert all(s["final"] < 5e-14 and s["iterations"] < 60 for s in ii["solves"].values()), ii
    for d, c in pairs:
        c.close(); d.close()
    whole.close(); gdev.close()
