"""Kernel times of the explicit step on an n^3 box under different cell orders (points and faces follow the cells):
natural (blockMesh), pencil:BY:BZ (rows of n cells grouped into BY x BZ bundles, so that the j+1 and k+1 neighbours of a row
are a few rows away instead of a plane away), morton.   usage: order_probe.py n order [order ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
import cases

n = int(sys.argv[1])
for order in sys.argv[2:]:
    mesh = q.PolyMesh.box(n, n, n)
    t0 = time.perf_counter()
    if order.startswith("pencil"):
        by, bz = (int(v) for v in order.split(":")[1:])
        lab = np.arange(n ** 3, dtype=np.int64)
        i, j, k = lab % n, (lab // n) % n, lab // (n * n)
        nJ = (n + by - 1) // by
        key = ((k // bz) * nJ + (j // by)) * (by * bz) + (k % bz) * by + (j % by)
        new = np.empty(n ** 3, dtype=np.int32)
        new[np.lexsort((i, key))] = np.arange(n ** 3, dtype=np.int32)      # bundles may be ragged at the upper edges
        del lab, i, j, k, key
        mesh.renumber(new)
    elif order == "morton":
        mesh.renumber(mesh.morton_order())
    tren = time.perf_counter() - t0
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(deltaT=0.1 / n / 1.3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    case.step(5)
    t0 = time.perf_counter(); case.step(20); case.info(); dt = (time.perf_counter() - t0) / 20
    case.timing(True); case.timing_reset(); case.step(10)
    kt = {nm: case.kernel_time(k)[0] / max(1, case.kernel_time(k)[1]) for nm, k in (("point", L.K_POINT), ("face", L.K_FACE), ("cell", L.K_CELL))}
    print(f"{order:14s} n={n} step {dt * 1e3:.3f} ms  {n ** 3 / dt / 1e6:.0f} Mcell-steps/s", {k: round(v, 3) for k, v in kt.items()},
          f"renumber {tren:.1f} s  minRho {case.info()['minRho']:.4f}", flush=True)
    case.close(); dev.close(); mesh.close()
