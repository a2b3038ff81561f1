"""per-kernel average times of the explicit step on an nx x ny x nz box (HIP events of the timing API)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q, cases
nx, ny, nz = [int(x) for x in sys.argv[1:4]]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
mesh = q.PolyMesh.box(nx, ny, nz, hi=(1.0, ny / nx, nz / nx))
dev = q.Device(mesh); case = q.QGDFoamCase(dev, q.default_options(deltaT=0.1 / nx / 1.3))
U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3)); case.set_fields(U, T, p)
case.step(10)
case.timing(True); case.timing_reset()
case.step(steps)
names = ["point", "face", "bface", "cell", "bc"]
out = []
for k, n in enumerate(names):
    ms, cnt = case.kernel_time(k)
    out.append(f"{n} {ms / max(cnt, 1):.4f}")
print(f"{nx}x{ny}x{nz}: " + "  ".join(out) + f"  min_rho {case.info()['minRho']:.4g}", flush=True)
