"""wall time of the resident QHDFoam step (qgd_qhd_case_step) on an n^3 box: usage  qhd_step_timing.py n [steps]
QGD_MG_GRAPH=1 replays the V-cycle as a captured hipGraph instead of launch by launch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
from test_qhd_case import cavity_bcs, options, initial
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mesh = q.PolyMesh.box(n, n, n)
dev = q.Device(mesh)
c = qhdfoam.QHDFoamCase(dev, options(deltaT=0.2 / n, pTol=1e-8, pMaxIter=400))
cavity_bcs(c, mesh)
c.set_fields(*initial(mesh))
c.step(2)
t0 = time.perf_counter(); c.step(steps); c.field("p")[:1]; dt = (time.perf_counter() - t0) / steps
info = c.info()
print(f"QHD {n}^3 graph={'on' if os.environ.get('QGD_MG_GRAPH') else 'off'}: {dt * 1e3:.2f} ms/step, {info}", flush=True)
