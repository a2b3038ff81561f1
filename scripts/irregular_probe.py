"""BASELINE config 5 stand-in ("irregular stencil stress"): the explicit QGDFoam step on an n^3 hex box whose vertices are
jittered, every `tri`-th quad split into triangles, and whose cells are relabelled at random within chunks of `chunk`
consecutive labels -- then the same mesh after the library's reverse Cuthill-McKee renumbering.
usage: irregular_probe.py n [tri_stride] [chunk] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q
import cases


def run(mesh, tag, steps):
    t0 = time.perf_counter()
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.05 / n / 1.3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    setup = time.perf_counter() - t0
    case.step(10)
    t0 = time.perf_counter()
    case.step(steps)
    wall = (time.perf_counter() - t0) / steps
    case.timing(True)
    case.timing_reset()
    case.step(20)
    names = ["point", "face", "bface", "cell", "bc"]
    parts = []
    for k, nm in enumerate(names):
        ms, cnt = case.kernel_time(k)
        parts.append(f"{nm} {ms / max(cnt, 1):.3f}")
    bw = int(np.abs(mesh.array("neighbour").astype(np.int64) - mesh.array("owner")[:mesh.nInternalFaces]).max())
    print(f"{tag:10s} {mesh.nCells / 1e6:6.2f} Mcells  bandwidth {bw:9d}  {1e3 * wall:7.3f} ms/step  {mesh.nCells / wall / 1e6:7.1f} Mcell-steps/s  "
          f"[{'  '.join(parts)}]  setup {setup:.1f} s  min_rho {case.info()['minRho']:.4f}  tiles {dev.face_tiles()}  fused {case.fused_info()}", flush=True)
    case.close()
    dev.close()


n = int(sys.argv[1])
tri = int(sys.argv[2]) if len(sys.argv) > 2 else 0
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 64 ** 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
mesh = q.PolyMesh.box(n, n, n)
mesh.jitter(0.2, seed=2024)
if tri:
    mesh.split_quads(tri)
run(mesh, "natural", steps)
rng = np.random.default_rng(7)
perm = np.arange(mesh.nCells, dtype=np.int32)
for a in range(0, mesh.nCells, chunk):
    b = min(a + chunk, mesh.nCells)
    perm[a:b] = a + rng.permutation(b - a)
t0 = time.perf_counter()
mesh.renumber(perm)
print(f"renumber: {time.perf_counter() - t0:.1f} s", flush=True)
run(mesh, "shuffled", steps)
t0 = time.perf_counter()
order = mesh.rcm_order()
t1 = time.perf_counter()
mesh.renumber(order)
print(f"rcm: order {t1 - t0:.1f} s, renumber {time.perf_counter() - t1:.1f} s", flush=True)
run(mesh, "rcm", steps)
t0 = time.perf_counter()
order = mesh.morton_order()
t1 = time.perf_counter()
mesh.renumber(order)
print(f"morton: order {t1 - t0:.1f} s, renumber {time.perf_counter() - t1:.1f} s", flush=True)
run(mesh, "morton", steps)
