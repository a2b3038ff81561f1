"""QHDFoam step on an anisotropic box (nx x ny x nz cells in the unit cube: aspect ratios nx:ny:nz): pressure iterations and step time
under the multigrid knobs of the environment.  usage: qhd_aniso_probe.py nx ny nz [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
from test_qhd_case import cavity_bcs, options, initial
nx, ny, nz = (int(a) for a in sys.argv[1:4]); steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
mesh = q.PolyMesh.box(nx, ny, nz)
dev = q.Device(mesh)
c = qhdfoam.QHDFoamCase(dev, options(deltaT=0.2 / max(nx, ny, nz), pTol=1e-8, pMaxIter=600))
cavity_bcs(c, mesh)
c.set_fields(*initial(mesh))
c.step(2)
t0 = time.perf_counter(); c.step(steps); c.sync(); dt = (time.perf_counter() - t0) / steps
i = c.info()
print(f"QHD {nx}x{ny}x{nz}: {dt * 1e3:.2f} ms/step, {i['pIterations']} iterations, levels {i['mgLevels']}, final residual {i['pFinalResidual']:.2e}", flush=True)
