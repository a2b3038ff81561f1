"""Pressure iterations of the QHDFoam case against the number of shards (every rank preconditions with the multigrid hierarchy of its own
rows): usage  qhd_shard_penalty.py n [worlds...]   -- an n^3 box cut into k-slabs, all shards resident on this GPU, 2 steps each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
from qgdsolver_amd.halo import LocalWorld, QhdStepper
from qhd_shards import box_slabs
from test_qhd_case import cavity_bcs, options, initial
n = int(sys.argv[1]); worlds = [int(w) for w in sys.argv[2:]] or [1, 2, 4, 8]
g = q.PolyMesh.box(n, n, n)
fields = initial(g)
opt = options(deltaT=0.2 / n, pTol=1e-8, pMaxIter=2000)
for world in worlds:
    if world == 1:
        dev = q.Device(g); c = qhdfoam.QHDFoamCase(dev, opt); cavity_bcs(c, g); c.set_fields(*fields)
        t0 = time.perf_counter(); c.step(2); c.sync(); dt = (time.perf_counter() - t0) / 2
        print(f"n={n} shards=1: {c.info()['pIterations']} iterations, {dt * 1e3:.1f} ms/step", flush=True)
        c.close(); dev.close(); continue
    shards = box_slabs(n, n, n, world)
    pairs = []
    for sh in shards:
        d = q.Device(sh["mesh"]); c = qhdfoam.QHDFoamCase(d, opt); cavity_bcs(c, sh["mesh"])
        cg = sh["cell_global"]; c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg]); pairs.append((d, c))
    cases = [c for _, c in pairs]
    st = QhdStepper(LocalWorld(cases, [sh["peers"] for sh in shards]))
    t0 = time.perf_counter(); st.step(2); [c.sync() for c in cases]; dt = (time.perf_counter() - t0) / 2
    print(f"n={n} shards={world}: {cases[0].info()['pIterations']} iterations, {dt * 1e3:.1f} ms/step (all shards on one GPU, in turn)", flush=True)
    for d, c in pairs: c.close(); d.close()
