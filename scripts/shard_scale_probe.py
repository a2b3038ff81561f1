"""Cell-range sharding at scale: n^3 box, Morton-renumbered, W shards driven in one process on one GPU, against the unsharded run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q, cases
from test_partition_gpu import run_device, run_sharded_device
n, world, steps = int(sys.argv[1]), int(sys.argv[2]), 5
g = q.PolyMesh.box(n, n, n)
t0 = time.perf_counter(); g.renumber(g.morton_order()); print(f"renumber {time.perf_counter()-t0:.1f} s", flush=True)
U, T, p = cases.box_initial_fields(g.array("C").reshape(-1, 3))
opt = dict(deltaT=0.1 / n / 1.3, mu=1e-3)
one = run_device(g, "GaussVolPoint", None, U, T, p, steps, **opt)
t0 = time.perf_counter()
shards = [g.shard(world, r) for r in range(world)]
print(f"shard x{world}: {time.perf_counter()-t0:.1f} s; cells per shard {[s.nCells for s in shards]}; peers {[list(s.array('haloPeer')) for s in shards]}", flush=True)
del shards
for ov in (False, True):
    got = run_sharded_device(g, world, "GaussVolPoint", None, U, T, p, steps, overlapped=ov, **opt)
    print("overlapped" if ov else "plain", {f: float(np.abs(got[f] - one[f]).max() / np.abs(one[f]).max()) for f in one}, flush=True)
