"""Where bench.py's setup_s goes at n^3 cells: mesh, device (QGD_SETUP_TIMING=1 prints the library's stages on stderr), case, initial fields.
    QGD_SETUP_TIMING=1 python scripts/setup_timing.py [n=400]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd.synthetic import box_initial_fields  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.perf_counter()
marks = []


def mark(what):
    global t0
    t = time.perf_counter()
    marks.append((what, t - t0))
    print(f"{what:52s} {t - t0:8.2f} s", flush=True)
    t0 = t


mesh = q.PolyMesh.box(n, n, n); mark("PolyMesh.box (points, faces, owner/neighbour)")
dev = q.Device(mesh); mark("Device (qgd_device_create: stages on stderr)")
case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.1 / n / 1.3)); mark("QGDFoamCase (records, flux buffers)")
C = mesh.array("C").reshape(-1, 3); mark("mesh.array('C')")
U, T, p = box_initial_fields(C); mark("box_initial_fields")
noise = np.random.Generator(np.random.MT19937(12345)).uniform(-1e-3, 1e-3, size=n * n * n); T = 1.0 + noise; mark("noise (MT19937)")
case.set_fields(U, T, p); mark("set_fields (upload + cellInit)")
case.step_phase(3); case.sync(); mark("first step")
print("total %.2f s" % sum(t for _, t in marks))
