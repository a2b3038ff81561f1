import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
import cases
from util import make_mesh, oracle_mesh_of, rel_err
from oracle import OracleCase
import test_case_parity_gpu as tc

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
mk, sch, bc, init, opt = tc.CASES[idx]
mesh, dev, gc, oc = tc.build_pair(mk, sch, bc, init, **opt)
names = tc.CELL_FIELDS + [n + ".boundary" for n in ("rho", "U", "p", "e", "c", "H", "muQGD")]
def report(tag):
    errs = {n: rel_err(gc.field(n), oc.field(n)) for n in names}
    bad = {k: "%.2e" % v for k, v in errs.items() if v > 1e-13}
    print(tag, bad)
report("init")
for s in range(4):
    gc.updateFluxes(); oc.updateFluxes()
    fe = {n: rel_err(gc.field(n), oc.field(n)) for n in tc.FACE_FIELDS}
    print(" fluxes", {k: "%.2e" % v for k, v in fe.items() if v > 1e-13})
    gc.step(1); oc.step(1)
    report("step%d" % (s + 1))
