"""one- and two-cell meshes through the implicit branch and the QHD case (no internal face at all / one)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
for dims in ((1, 1, 1), (2, 1, 1), (3, 2, 1)):
    mesh = q.PolyMesh.box(*dims)
    n = mesh.nCells
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(deltaT=1e-3, mu=1e-3, implicitDiffusion=1))
    case.set_fields(np.zeros((n, 3)), np.ones(n), np.ones(n))
    case.step(3)
    print(dims, "implicit rho", case.field("rho"), case.implicit_info()["solves"]["e"])
    qc = qhdfoam.QHDFoamCase(dev, qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=1.0, rho0=1.0, mu=1e-2, Pr=0.71,
                                                     beta=3e-3, g=(0.0, -9.81, 0.0), deltaT=1e-3, pTol=1e-10, pMaxIter=200, pRefCell=0, pRefValue=0.0))
    for ip in range(mesh.nPatches):
        qc.set_bc(ip, U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("zeroGradient", None))
    qc.set_fields(np.zeros((n, 3)), np.full(n, 300.0), np.zeros(n))
    qc.step(3)
    print(dims, "qhd T", qc.field("T"), qc.info()["pIterations"])
    qc.close(); case.close(); dev.close()
print("ok")
