"""wall time of qgd_qhd_pressure on an n^3 box (host-pointer entry: includes PCIe for 3 nF + nC doubles each way)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
n = int(sys.argv[1])
mesh = q.PolyMesh.box(n, n, n)
dev = q.Device(mesh)
rng = np.random.default_rng(1)
nF = mesh.nFaces
phiu = 1e-2 * rng.standard_normal(nF); phiwo = 1e-3 * rng.standard_normal(nF); tbr = 1e-3 * (1 + 0.3 * rng.random(nF))
kinds = ["fixedValue"] + ["zeroGradient"] * 5
pb = np.ones(mesh.nBoundaryFaces)
for tol, it in ((1e-30, 10), (1e-30, 110), (1e-8, 100000)):
    t0 = time.perf_counter()
    p, phi, info = qhdfoam.pEqn(dev, phiu, phiwo, tbr, np.zeros(mesh.nCells), kinds, pb, None, tolerance=tol, maxIter=it)
    print(f"n={n} maxIter={it} tol={tol:g}: {info['iterations']} iterations, residual {info['finalResidual']:.2e}, {time.perf_counter() - t0:.3f} s", flush=True)
