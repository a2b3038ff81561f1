import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_fused_step_gpu as t
from test_config5_gpu import c5_mesh
mesh = c5_mesh(16, 8 ** 3, poly=True)
a, _ = t.run(mesh, 1, False, deltaT=0.0025)
b, ib = t.run(mesh, 1, True, deltaT=0.0025)
print(os.environ.get("QGD_AMD_LIB"), ib["fused"], {k: float(np.abs(a[k] - b[k]).max()) for k in ("rho", "U", "p", "e")}, int((a["rho"] != b["rho"]).sum()), "cells differ of", a["rho"].size)
import qgdsolver_amd as q
tri = q.PolyMesh.box(9, 7, 5); tri.jitter(0.15, seed=3)
a, _ = t.run(tri, 1, False, deltaT=0.0025)
b, ib = t.run(tri, 1, True, deltaT=0.0025)
print("jittered hexahedra only:", int((a["rho"] != b["rho"]).sum()), "cells differ of", a["rho"].size)
