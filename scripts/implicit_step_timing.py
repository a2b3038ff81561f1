"""wall time of the QGDFoam step with implicitDiffusion true (the reference's default branch) on an n^3 box:  implicit_step_timing.py n [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
import cases
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mesh = q.PolyMesh.box(n, n, n)
dev = q.Device(mesh)
for impl in ((1,) if os.environ.get('QGD_IMPL_ONLY') else (0, 1)):
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.1 / n / 1.3, implicitDiffusion=impl, mu=1e-3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    case.step(2)
    t0 = time.perf_counter(); case.step(steps); case.field("rho")[:1]; dt = (time.perf_counter() - t0) / steps
    print(f"n={n} implicitDiffusion={impl}: {dt * 1e3:.2f} ms/step  {n ** 3 / dt / 1e6:.0f} Mcell-steps/s  {case.info()}", flush=True)
    case.close()
