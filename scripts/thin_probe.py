import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qgdsolver_amd as q, cases
nx, ny, nz = [int(x) for x in sys.argv[1:4]]
mesh = q.PolyMesh.box(nx, ny, nz, hi=(1.0, ny / nx, nz / nx))
dev = q.Device(mesh); case = q.QGDFoamCase(dev, q.default_options(deltaT=0.1 / nx / 1.3))
U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3)); case.set_fields(U, T, p)
case.step(6)
