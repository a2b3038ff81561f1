import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_case_parity_gpu as tc
from util import rel_err
idx = int(sys.argv[1])
mk, sch, bc, init, opt = tc.CASES[idx]
mesh, dev, gc, oc = tc.build_pair(mk, sch, bc, init, **opt)
def show(tag, names):
    e = {n: rel_err(gc.field(n), oc.field(n)) for n in names}
    print(tag, {k: "%.1e" % v for k, v in e.items() if v > 1e-13})
gc.updateFluxes(); oc.updateFluxes()
show("flux0", tc.FACE_FIELDS)
show("bnd0", [n + ".boundary" for n in ("rho", "U", "p", "e", "c", "H", "muQGD")])
for chunk in (1, 4, 20):
    gc.step(chunk); oc.step(chunk)
    show("steps+%d" % chunk, ["rho", "U", "p", "e", "rhoU", "rhoE"])
gc.updateFluxes(); oc.updateFluxes()
show("fluxN", tc.FACE_FIELDS)
a, b = gc.field("phiQ"), oc.field("phiQ")
i = np.argmax(np.abs(a - b)); print("worst face", i, "internal" if i < mesh.nInternalFaces else "boundary", a[i], b[i], "max|phiQ|", np.abs(b).max(), "face size", np.diff(mesh.array("faceOffsets"))[i])
