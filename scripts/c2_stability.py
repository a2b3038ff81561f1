import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q, cases
mesh = q.PolyMesh.forward_step(600, 200, 120, 40)
n = mesh.nCells
U = np.zeros((n, 3)); U[:, 0] = 3.0
for stencil in ("leastSquares", "GaussVolPoint"):
    for dt, alpha in ((1e-4, 0.5), (5e-5, 0.5), (1e-4, 0.8), (1e-4, 1.0), (2e-4, 1.0)):
        dev = q.Device(mesh)
        gc = q.QGDFoamCase(dev, q.default_options(stencil=stencil, deltaT=dt, alphaQGD=alpha))
        cases.forward_step_bcs(gc)
        gc.set_fields(U, np.ones(n), np.ones(n))
        hist = []
        for k in range(20):
            gc.step(100)
            i = gc.info()
            hist.append(i["minRho"])
            if not (i["minRho"] > 0): break
        print(stencil, dt, alpha, " ".join(f"{h:.3g}" for h in hist), flush=True)
        gc.close(); dev.close()
