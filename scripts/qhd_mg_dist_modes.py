"""The three ways a sharded pressure solve builds its multigrid hierarchy (QGD_MG_DIST = 1: the global level-0 matrix gathered and coarsened on
every rank | 2: level 0 coarsened per rank, only the level-1 matrix gathered | 0: rank-local hierarchies): pressure iterations and set-up time
(the first step, which builds the hierarchy) of an n^3 box cut into `world` k-slabs, all shards resident on this GPU, driven in turn.
    python scripts/qhd_mg_dist_modes.py [n=128] [world=8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd import qhdfoam  # noqa: E402
from qgdsolver_amd.halo import LocalWorld, QhdStepper  # noqa: E402
from qhd_shards import box_slabs  # noqa: E402
from test_qhd_case import cavity_bcs, options, initial  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = q.PolyMesh.box(n, n, n)
fields = initial(g)
opt = options(deltaT=0.2 / n, pTol=1e-8, pMaxIter=2000)
dev = q.Device(g, fused_tables=False); c = qhdfoam.QHDFoamCase(dev, opt); cavity_bcs(c, g); c.set_fields(*fields)
t0 = time.perf_counter(); c.step(1); c.sync(); t1 = time.perf_counter() - t0
c.step(2)
print(f"n={n} unsharded: {c.info()['pIterations']} iterations in step 3; first step (hierarchy of {g.nCells} rows on the host) {t1:.2f} s", flush=True)
c.close(); dev.close()
for mode in ("1", "2", "0"):
    os.environ["QGD_MG_DIST"] = mode
    shards = box_slabs(n, n, n, world)
    pairs = []
    for sh in shards:
        d = q.Device(sh["mesh"], fused_tables=False); c = qhdfoam.QHDFoamCase(d, opt); cavity_bcs(c, sh["mesh"])
        cg = sh["cell_global"]; c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg]); pairs.append((d, c))
    cases = [c for _, c in pairs]
    st = QhdStepper(LocalWorld(cases, [sh["peers"] for sh in shards]))
    t0 = time.perf_counter(); st.step(1); [c.sync() for c in cases]; t1 = time.perf_counter() - t0
    st.step(2); [c.sync() for c in cases]
    print(f"n={n} shards={world} QGD_MG_DIST={mode}: {cases[0].info()['pIterations']} iterations in step 3, {cases[0].info()['mgLevels']} levels; "
          f"first step of ALL {world} shards in turn on this host {t1:.2f} s", flush=True)
    for d, c in pairs:
        c.close(); d.close()
