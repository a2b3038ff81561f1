import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
import test_case_parity_gpu as tc
idx = int(sys.argv[1]); fld = sys.argv[2]
mk, sch, bc, init, opt = tc.CASES[idx]
mesh, dev, gc, oc = tc.build_pair(mk, sch, bc, init, **opt)
gc.updateFluxes(); oc.updateFluxes()
a = gc.field(fld); b = oc.field(fld)
d = np.abs(a - b).reshape(len(a), -1).max(axis=1)
bad = np.argsort(-d)[:8]
nIF = mesh.nInternalFaces
ps = mesh.array("patchStart"); 
for f in bad:
    pid = -1 if f < nIF else int(np.searchsorted(ps, f, side="right") - 1)
    print(f, "internal" if f < nIF else "patch %d" % pid, d[f], a[f], b[f])
for n in ("p", "rho", "e"):
    print(n, "bnd diff", np.abs(gc.field(n + ".boundary") - oc.field(n + ".boundary")).max())
