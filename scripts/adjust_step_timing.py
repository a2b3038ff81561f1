"""The explicit step under Courant-number control (adjustTimeStep) on an n^3 box: the three kernels (vertex values, faces with the Courant
partials, cells) against the cell blocks (fusedFaceCellKernel<..., ADJ>: every block up to its flux sums + Courant partials, faceReduce, deltaT,
cellFinishKernel), same process, same box.    python scripts/adjust_step_timing.py [n=400] [steps=50]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd.synthetic import box_initial_fields  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mesh = q.PolyMesh.box(n, n, n)
U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
for rep in range(2):
    for tables in (False, True):
        t0 = time.perf_counter()
        dev = q.Device(mesh, fused_tables=tables)
        case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.05 / n / 1.3, adjustTimeStep=1, maxCo=0.1, maxDeltaT=1.0))
        case.set_fields(U, T, p)
        setup = time.perf_counter() - t0
        fi = case.fused_info()
        case.step(10)
        t0 = time.perf_counter()
        case.step(steps)
        ms = (time.perf_counter() - t0) / steps * 1e3
        i = case.info()
        print(json.dumps({"n": n, "path": "cell blocks (2 launches)" if fi["fusedAdjust"] else "three kernels", "ms_per_step": round(ms, 4),
                          "Mcell_steps_per_s": round(mesh.nCells / ms / 1e3, 1), "deltaT": i["deltaT"], "CoNum": i["CoNum"], "setup_s": round(setup, 1),
                          "device_GB": round(case.device_bytes() / 1e9, 1)}), flush=True)
        case.close(); dev.close()
