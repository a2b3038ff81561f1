"""Launch-bound regime: time per step of the C2 forwardStep case (100 800 cells) and of small boxes."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import qgdsolver_amd as q
import cases

def run(mesh, stencil, bc_fn, init, dt, steps=2000):
    dev = q.Device(mesh)
    gc = q.QGDFoamCase(dev, q.default_options(stencil=stencil, deltaT=dt))
    if bc_fn: bc_fn(gc)
    gc.set_fields(*init)
    gc.step(200); gc.sync()
    t0 = time.perf_counter(); gc.step(steps); gc.sync(); t = time.perf_counter() - t0
    print(f"{mesh.nCells:9d} cells {stencil:14s} {1e6*t/steps:8.1f} us/step  {mesh.nCells*steps/t/1e6:8.1f} Mcell-steps/s", flush=True)
    gc.close(); dev.close()

mesh = q.PolyMesh.forward_step(600, 200, 120, 40)
n = mesh.nCells
U = np.zeros((n, 3)); U[:, 0] = 3.0
for st in ("leastSquares", "GaussVolPoint", "reduced"):
    run(mesh, st, cases.forward_step_bcs, (U, np.ones(n), np.ones(n)), 5e-4)
for e in (16, 32, 64, 100):
    m = q.PolyMesh.box(e, e, e)
    run(m, "GaussVolPoint", None, cases.box_initial_fields(m.array("C").reshape(-1, 3)), 0.1 / e / 1.3)
