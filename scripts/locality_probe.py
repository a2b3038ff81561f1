"""Kernel times on boxes of equal cell count but different plane sizes: an upper bound of what a cache-blocked
internal ordering can give (small planes == short reuse distances)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
import cases
for dims in [(200, 200, 200), (64, 64, 1953), (32, 32, 7812), (16, 16, 31250)]:
    nx, ny, nz = dims
    mesh = q.PolyMesh.box(nx, ny, nz, hi=(1.0, ny / nx, nz / nx))
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(deltaT=0.1 / nx / 1.3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    case.step(5)
    case.timing(True); case.timing_reset()
    t0 = time.perf_counter(); case.step(20); dt = (time.perf_counter() - t0) / 20
    kt = {n: case.kernel_time(k)[0] / max(1, case.kernel_time(k)[1]) for n, k in (("point", L.K_POINT), ("face", L.K_FACE), ("cell", L.K_CELL))}
    print(dims, "cells %.2fM step %.3f ms" % (mesh.nCells / 1e6, dt * 1e3), {k: round(v, 3) for k, v in kt.items()}, flush=True)
    case.close(); dev.close(); mesh.close()
