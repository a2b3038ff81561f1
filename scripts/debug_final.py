import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_case_parity_gpu as tc
from util import rel_err
idx = int(sys.argv[1])
mk, sch, bc, init, opt = tc.CASES[idx]
mesh, dev, gc, oc = tc.build_pair(mk, sch, bc, init, **opt)
for chunk in (1, 4, 20):
    gc.step(chunk); oc.step(chunk)
    print("steps+%d" % chunk, {n: "%.1e" % rel_err(gc.field(n), oc.field(n)) for n in ("rho", "U", "p", "e")})
gc.updateFluxes(); oc.updateFluxes()
print({n: "%.1e" % rel_err(gc.field(n), oc.field(n)) for n in tc.FACE_FIELDS})
a, b = gc.field("phiQ"), oc.field("phiQ")
i = np.argmax(np.abs(a - b)); print("worst face", i, "internal" if i < mesh.nInternalFaces else "boundary", a[i], b[i], "max|phiQ|", np.abs(b).max())
ge, oe = gc.field("gradef"), oc.field("gradef"); print("gradef at face", ge[i], oe[i])
