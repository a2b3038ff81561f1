import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import fvsc
import cases
from util import make_mesh, oracle_mesh_of, rel_err
import test_fvsc_parity_gpu as tf
for mk, sch in tf.MESH_SCHEMES:
    mesh = make_mesh(mk); om = oracle_mesh_of(mesh)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": sch}})
    res = {}
    for op, nc in tf.OPS:
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, 3)
        rc, ref = om.fvsc(sch, op, cell, bnd)
        vf = q.volField("f", cell, bnd)
        got = fvsc.grad(dev, vf) if op.startswith("grad") else fvsc.div(dev, vf)
        nIF = mesh.nInternalFaces
        res[op] = "%.1e/%.1e" % (rel_err(got[:nIF], ref[:nIF]), rel_err(got[nIF:], ref[nIF:]))
    print(mk, sch, res)
    dev.close()
