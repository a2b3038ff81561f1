"""QHDFoam on cell-range shards: the step as phases with reductions of the control block and three kinds of
halo message between them (include/qgd_amd.h "the QHD case on a cell-range shard", qgdsolver_amd.halo.QhdStepper).

CPU: the oracle's phases on an UNSHARDED mesh reproduce its monolithic step; 2-3 shards of the oracle in one process
(LocalWorld: box slabs and cell ranges of a renumbered polygonal mesh, a reference cell owned by one shard, a fixedValue
patch present on one shard only) against the unsharded oracle; the same over gloo with one rank per shard (DistWorld).
GPU: several HIP shards on one device against the unsharded HIP run and the oracle; the 16 M-cell config-5 mesh cut 8-way."""
import os
import subprocess
import sys

import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
from qgdsolver_amd.halo import LocalWorld, QhdStepper

from oracle import OracleQhdCase
from qhd_shards import box_slabs, gather, make_oracle_shard_case, oracle_shard_mesh, range_shards
from test_qhd_case import cavity_bcs, initial, options
from util import make_mesh, oracle_mesh_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = (("U", 3), ("T", 1), ("p", 1))


def perturbed(mesh, seed=3):
    U, T, p = initial(mesh)
    U = U + 1e-2 * np.random.default_rng(seed).standard_normal(U.shape)
    if mesh.nGeometricD == 2:
        U[:, 2] = 0.0
    return U, T, p


def open_box_bcs(case, mesh):
    """a fixedValue pressure patch (zMax): no reference level; on an inner / lower k-slab that patch does not exist"""
    cavity_bcs(case, mesh)
    case.set_bc(5, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 0.0))


def run_unsharded_oracle(mesh, opt, bc_fn, fields, steps, phases=False):
    oc = OracleQhdCase(oracle_mesh_of(mesh), opt)
    bc_fn(oc, mesh)
    oc.set_fields(*fields)
    if phases:
        QhdStepper(LocalWorld([oc], [[]])).step(steps)
    else:
        oc.step(steps)
    return oc


@pytest.mark.parametrize("kind,stencil", [("box654_jitter", "GaussVolPoint"), ("plane2d_jitter", "leastSquares")])
def test_oracle_phases_are_the_oracle_step(kind, stencil):
    mesh = make_mesh(kind)
    opt = options(stencil, deltaT=1e-3, precond=0)
    fields = perturbed(mesh)
    a = run_unsharded_oracle(mesh, opt, cavity_bcs, fields, 6)
    b = run_unsharded_oracle(mesh, opt, cavity_bcs, fields, 6, phases=True)
    for f in ("U", "T", "p", "phi", "p.boundary", "U.boundary"):
        ra, rb = a.field(f), b.field(f)
        assert np.abs(ra - rb).max() <= 1e-11 * max(np.abs(ra).max(), 1e-300), f
    assert a.info()["pIterations"] == b.info()["pIterations"] > 0 and b.info()["steps"] == 6


CASES = [("slabs", 2, cavity_bcs, 7), ("slabs", 3, cavity_bcs, 300), ("slabs", 3, open_box_bcs, 0), ("ranges", 3, cavity_bcs, 111),
         ("ranges", 2, open_box_bcs, 0)]


@pytest.mark.parametrize("cut,world,bc_fn,ref_cell", CASES)
def test_sharded_oracle_matches_unsharded(cut, world, bc_fn, ref_cell):
    if cut == "slabs":
        g = q.PolyMesh.box(6, 5, 12)
        shards = box_slabs(6, 5, 12, world)
    else:
        g = make_mesh("box654_poly")
        g.renumber(np.random.default_rng(4).permutation(g.nCells).astype(np.int32))
        g.renumber(g.rcm_order())
        shards = range_shards(g, world)
    opt = options("GaussVolPoint", deltaT=1e-3, precond=0, pRefCell=ref_cell, pRefValue=0.25)
    fields = perturbed(g)
    steps = 5
    ref = run_unsharded_oracle(g, opt, bc_fn, fields, steps)
    need_ref = bc_fn is cavity_bcs
    cases = [make_oracle_shard_case(sh, opt, bc_fn, fields, ref_cell, need_ref) for sh in shards]
    QhdStepper(LocalWorld(cases, [sh["peers"] for sh in shards])).step(steps)
    for f, nc in FIELDS:
        got, want = gather(shards, cases, f, g.nCells, nc), ref.field(f)
        assert np.abs(got - want).max() <= 1e-9 * max(np.abs(want).max(), 1e-300), (f, np.abs(got - want).max())
    if need_ref:
        assert abs(gather(shards, cases, "p", g.nCells)[ref_cell] - 0.25) <= 1e-12     # the shifted reference level, whoever owns the cell
    its = [c.info()["pIterations"] for c in cases]
    assert len(set(its)) == 1 and abs(its[0] - ref.info()["pIterations"]) <= 2          # Jacobi-PCG is the same iteration on any cut


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_oracle_over_gloo(tmp_path, world):
    steps = 4
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + world), os.path.join(ROOT, "tests", "qhd_halo_worker.py"), str(tmp_path), str(steps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = make_mesh("box654_poly")
    g.renumber(g.rcm_order())
    opt = options("GaussVolPoint", deltaT=1e-3, precond=0, pRefCell=17, pRefValue=0.0)
    ref = run_unsharded_oracle(g, opt, cavity_bcs, perturbed(g), steps)
    covered = 0
    for rank in range(world):
        d = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        covered += d["cells"].size
        for f, _ in FIELDS:
            want = ref.field(f)[d["cells"]]
            assert np.abs(d[f] - want).max() <= 1e-9 * max(np.abs(ref.field(f)).max(), 1e-300), (rank, f)
        assert int(d["iterations"]) > 0
    assert covered == g.nCells


# ---- GPU: HIP shards ----------------------------------------------------------------------------------------------------
def make_device_shard_case(sh, opt, bc_fn, fields):
    dev = q.Device(sh["mesh"])
    c = qhdfoam.QHDFoamCase(dev, opt)     # pRefCell of the options is a label of the unsharded mesh: the library maps it
    bc_fn(c, sh["mesh"])
    cg = sh["cell_global"]
    c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg])
    return dev, c


GPU_CUTS = [("slabs", 3, cavity_bcs, 300, 1), ("slabs", 2, open_box_bcs, 0, 1), ("ranges", 3, cavity_bcs, 111, 1), ("ranges", 4, cavity_bcs, 5, 0),
            ("tri-ranges", 3, cavity_bcs, 40, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("cut,world,bc_fn,ref_cell,precond", GPU_CUTS)
def test_device_shards_match_unsharded_device_and_oracle(cut, world, bc_fn, ref_cell, precond):
    """several HIP shards resident on the one GPU of the box, stepped in lockstep by the same QhdStepper the ranks of a real
    run use (LocalWorld: messages device-to-device through the pack / unpack kernels, reductions through the control blocks)"""
    if cut == "slabs":
        g = q.PolyMesh.box(10, 9, 12)
        shards = box_slabs(10, 9, 12, world)
    else:
        g = make_mesh("box654_poly" if cut == "ranges" else "box654_tri")
        g.renumber(np.random.default_rng(4).permutation(g.nCells).astype(np.int32))
        g.renumber(g.morton_order())
        shards = range_shards(g, world)
    opt = options("GaussVolPoint", deltaT=1e-3, precond=precond, pRefCell=ref_cell, pRefValue=0.25, pTol=1e-12)
    fields = perturbed(g)
    steps = 6
    ref = run_unsharded_oracle(g, options("GaussVolPoint", deltaT=1e-3, precond=0, pRefCell=ref_cell, pRefValue=0.25, pTol=1e-13), bc_fn, fields, steps)
    gdev = q.Device(g)
    whole = qhdfoam.QHDFoamCase(gdev, opt)
    bc_fn(whole, g)
    whole.set_fields(*fields)
    whole.step(steps)
    pairs = [make_device_shard_case(sh, opt, bc_fn, fields) for sh in shards]
    cases = [c for _, c in pairs]
    QhdStepper(LocalWorld(cases, [sh["peers"] for sh in shards])).step(steps)
    for f, nc in FIELDS:
        got = gather(shards, cases, f, g.nCells, nc)
        for tag, want in (("unsharded device", whole.field(f)), ("oracle", ref.field(f))):
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
            assert err <= 1e-8, (cut, f, tag, err)
    infos = [c.info() for c in cases]
    assert all(i["steps"] == steps and i["pFinalResidual"] < 1e-12 for i in infos), infos
    assert len({i["pIterations"] for i in infos}) == 1
    # the block (additive Schwarz) preconditioner costs iterations; on meshes this small (a few hundred cells per shard, one
    # multigrid level) the bound only guards against a preconditioner that stopped working
    assert infos[0]["pIterations"] <= 6 * whole.info()["pIterations"] + 20, (infos[0], whole.info())
    for d, c in pairs:
        c.close(); d.close()
    whole.close(); gdev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cut,world", [("slabs", 3), ("ranges", 4)])
def test_multigrid_hierarchy_spans_the_ranks(cut, world, monkeypatch):
    """Meshes large enough for a real hierarchy (> 1024 cells: below that the ranks keep their own): the first step gathers the global
    matrix (two all-reduces through qgd_qhd_case_pending), level 0 stays distributed, the coarse levels are replicated -- the sharded
    solve then needs the unsharded solve's iterations, where rank-local hierarchies (QGD_MG_DIST=0) need several times as many; the
    fields agree with the unsharded device case and the oracle either way."""
    if cut == "slabs":
        g = q.PolyMesh.box(16, 14, 18)
        make = lambda: box_slabs(16, 14, 18, world)
    else:
        from qhd_halo_worker_gpu import spanning_test_mesh
        g = spanning_test_mesh()
        make = lambda: range_shards(g, world)
    opt = options("GaussVolPoint", deltaT=1e-3, pRefCell=g.nCells // 2, pRefValue=0.1, pTol=1e-11)
    fields = perturbed(g)
    steps = 3
    ref = run_unsharded_oracle(g, options("GaussVolPoint", deltaT=1e-3, precond=0, pRefCell=g.nCells // 2, pRefValue=0.1, pTol=1e-13), cavity_bcs, fields, steps)
    gdev = q.Device(g)
    whole = qhdfoam.QHDFoamCase(gdev, opt)
    cavity_bcs(whole, g)
    whole.set_fields(*fields)
    whole.step(steps)
    assert whole.info()["mgLevels"] >= 2
    iters = {}
    # "cap": the hierarchy that spans the ranks replicates the global matrix on every rank, so above QGD_MG_DIST_MAX_CELLS (default
    # 20 M cells) a solve keeps the rank-local hierarchy and says so on stderr -- same answer, the iterations of QGD_MG_DIST=0
    # "2": level 0 coarsened PER RANK (aggregates and prolongator from the rank's own cells, cells next to a cut unsmoothed), only the
    # level-1 matrix gathered and replicated: nothing of the global level-0 matrix on any rank -- within a few iterations of "1"
    # "cap2": a mesh above QGD_MG_DIST_MAX_CELLS but within eight times it takes the per-rank coarsening by itself; "cap": above that, rank-local
    for dist in ("1", "2", "0", "cap2", "cap"):
        monkeypatch.setenv("QGD_MG_DIST", {"0": "0", "2": "2"}.get(dist, "1"))
        monkeypatch.delenv("QGD_MG_DIST_MAX_CELLS", raising=False)
        if dist in ("cap", "cap2"):
            monkeypatch.setenv("QGD_MG_DIST_MAX_CELLS", "100" if dist == "cap" else "1000")
        shards = make()
        pairs = [make_device_shard_case(sh, opt, cavity_bcs, fields) for sh in shards]
        cases = [c for _, c in pairs]
        QhdStepper(LocalWorld(cases, [sh["peers"] for sh in shards])).step(steps)
        for f, nc in FIELDS:
            got = gather(shards, cases, f, g.nCells, nc)
            for tag, want in (("unsharded device", whole.field(f)), ("oracle", ref.field(f))):
                err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
                assert err <= 1e-7, (cut, dist, f, tag, err)
        its = {c.info()["pIterations"] for c in cases}
        assert len(its) == 1
        iters[dist] = its.pop()
        for d, c in pairs:
            c.close(); d.close()
    assert iters["1"] <= whole.info()["pIterations"] + 2, (iters, whole.info())
    assert iters["2"] <= iters["1"] + 3 and iters["2"] < iters["0"], iters
    assert iters["0"] > iters["1"], iters
    assert iters["cap"] == iters["0"] and iters["cap2"] == iters["2"], iters
    whole.close(); gdev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_device_shards_over_gloo(tmp_path, world):
    """the sharded device case as separate processes (all on the one GPU of the box, gloo with host-staged buffers): halo messages,
    the all-reduced PCG scalars and the comm points of the multigrid hierarchy that spans the ranks through DistWorld"""
    from qhd_halo_worker_gpu import spanning_test_mesh
    steps = 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29570 + world), os.path.join(ROOT, "tests", "qhd_halo_worker_gpu.py"), str(tmp_path), str(steps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = spanning_test_mesh()
    opt = options("GaussVolPoint", deltaT=1e-3, pRefCell=g.nCells // 2, pRefValue=0.1, pTol=1e-11)
    fields = perturbed(g)
    gdev = q.Device(g)
    whole = qhdfoam.QHDFoamCase(gdev, opt)
    cavity_bcs(whole, g)
    whole.set_fields(*fields)
    whole.step(steps)
    covered = 0
    for rank in range(world):
        d = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        covered += d["cells"].size
        for f, _ in FIELDS:
            want = whole.field(f)[d["cells"]]
            assert np.abs(d[f] - want).max() <= 1e-7 * max(np.abs(whole.field(f)).max(), 1e-300), (rank, f)
        assert 0 < int(d["iterations"]) <= whole.info()["pIterations"] + 2, (int(d["iterations"]), whole.info())
        assert int(d["levels"]) == whole.info()["mgLevels"]      # the same global hierarchy on every rank
    assert covered == g.nCells
    whole.close(); gdev.close()


@pytest.mark.gpu
def test_native_step_on_one_rank_is_the_plain_step():
    """qgd_qhd_case_step_sharded over a one-rank communicator on an unsharded mesh: the library's own loop (ncclAllReduce and
    exchanges degenerate to nothing) gives the plain step bit for bit"""
    from qgdsolver_amd.halo import NativeComm
    from qgdsolver_amd import _lib as L
    import ctypes as C
    mesh = make_mesh("box654_jitter")
    opt = options("GaussVolPoint", deltaT=1e-3)
    dev = q.Device(mesh)
    a, b = qhdfoam.QHDFoamCase(dev, opt), qhdfoam.QHDFoamCase(dev, opt)
    fields = perturbed(mesh)
    for c in (a, b):
        cavity_bcs(c, mesh)
        c.set_fields(*fields)
    a.step(4)
    comm = NativeComm(0)
    L.check(L.lib.qgd_qhd_case_step_sharded(b._h, comm._h, None, 0, 4), "qgd_qhd_case_step_sharded")
    for f in ("U", "T", "p", "phi"):
        assert np.array_equal(a.field(f), b.field(f)), f
    assert a.info()["pIterations"] == b.info()["pIterations"] > 0
    a.close(); b.close(); dev.close(); comm.close()


@pytest.mark.gpu
def test_sharded_case_refuses_the_plain_step():
    sh = box_slabs(6, 5, 12, 2)[0]
    dev = q.Device(sh["mesh"])
    c = qhdfoam.QHDFoamCase(dev, options("GaussVolPoint"))
    cavity_bcs(c, sh["mesh"])
    n = sh["mesh"].nCells
    c.set_fields(np.zeros((n, 3)), np.full(n, 300.0), np.zeros(n))
    with pytest.raises(q.QgdError):
        c.step(1)
    c.close(); dev.close()
