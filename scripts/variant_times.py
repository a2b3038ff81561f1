"""per-kernel times of the explicit step for the other stencil variants (SURVEY 8d "other variants"):
reduced in 3-D, leastSquares / GaussVolPoint / reduced on a large one-cell-thick 2-D mesh"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
import cases
G, E = L.PATCH_GENERIC, L.PATCH_EMPTY

def run(mesh, stencil, tag, h):
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil=stencil, deltaT=0.1 * h / 1.3))
    C = mesh.array("C").reshape(-1, 3)
    U, T, p = cases.box_initial_fields(C)
    if mesh.nGeometricD == 2:
        U[:, 2] = 0.0
        for patch in (4, 5):
            case.set_bc(patch, U=("none", None), T=("none", None), p=("none", None))
    case.set_fields(U, T, p)
    case.step(10)
    case.timing(True); case.timing_reset()
    case.step(30)
    parts = []
    tot = 0.0
    for k, nm in enumerate(["point", "face", "bface", "cell", "bc"]):
        ms, cnt = case.kernel_time(k)
        parts.append(f"{nm} {ms / max(cnt, 1):.3f}")
        tot += ms / 30
    print(f"{tag:28s} {mesh.nCells / 1e6:5.2f} Mcells  kernels {tot:7.3f} ms/step  {mesh.nCells / tot / 1e3:7.1f} Mcell-steps/s  [{'  '.join(parts)}]  min_rho {case.info()['minRho']:.3f}", flush=True)
    case.close(); dev.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m3 = q.PolyMesh.box(n, n, n)
for st in ("GaussVolPoint", "reduced"):
    run(m3, st, f"3-D {n}^3 {st}", 1.0 / n)
m3.close()
nx = int(round((n ** 3) ** 0.5))
m2 = q.PolyMesh.box(nx, nx, 1, hi=(1.0, 1.0, 1.0 / nx), patch_types=[G, G, G, G, E, E])
for st in ("leastSquares", "GaussVolPoint", "reduced"):
    run(m2, st, f"2-D {nx}^2 {st}", 1.0 / nx)
