"""Host-side mirror of the QHDFoam flux-assembly fragments over the C-ABI (qgd_qhd_fluxes).

``updateFluxes(...)`` covers QHDFoam/updateFields.H L36-73 + updateFluxes.H L33-38 (call it without p/phi before the
pressure equation) and the flux parts of QHDUEqn.H L36-43 / QHDTEqn.H L65-66 (call it again with p and phi).
The pressure Poisson solve itself (QHDpEqn.H) is not on this path.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .qgdfoam import STENCIL_IDS


class _In(C.Structure):
    _fields_ = [(n, L.c_double_p) for n in ("U", "Ub", "T", "Tb", "p", "pb", "rho", "rhob", "tauQGDf", "phi")] + \
               [("beta", C.c_double), ("g", C.c_double * 3)]


class _Out(C.Structure):
    _fields_ = [(n, L.c_double_p) for n in ("gradUf", "gradTf", "phiu", "phiwo", "taubyrhof", "gradPf", "Wf", "phiUf", "phiTf",
                                            "phiTauTReg")]


L.lib.qgd_qhd_fluxes.restype = C.c_int
L.lib.qgd_qhd_fluxes.argtypes = [L.handle, C.c_int, C.POINTER(_In), C.POINTER(_Out)]

NCOMP = dict(gradUf=9, gradTf=3, phiu=1, phiwo=1, taubyrhof=1, gradPf=3, Wf=3, phiUf=3, phiTf=1, phiTauTReg=1)
NEED_P = {"gradPf", "Wf", "phiUf"}
NEED_PHI = {"phiUf", "phiTf"}


def updateFluxes(dev, scheme, U, T, rho, tauQGDf, beta, g, p=None, phi=None, struct_types=(_In, _Out), call=None):
    """U, T, rho (and p): (internal, boundary) pairs; returns a dict of face fields."""
    In, Out = struct_types
    m = dev.mesh
    keep = []

    def ptr(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.size == 0:
            a = np.zeros(1)
        keep.append(a)
        return a.ctypes.data_as(L.c_double_p)

    i = In()
    i.U, i.Ub = ptr(U[0]), ptr(U[1])
    i.T, i.Tb = ptr(T[0]), ptr(T[1])
    i.rho, i.rhob = ptr(rho[0]), ptr(rho[1])
    i.tauQGDf = ptr(tauQGDf)
    if p is not None:
        i.p, i.pb = ptr(p[0]), ptr(p[1])
    if phi is not None:
        i.phi = ptr(phi)
    i.beta = float(beta)
    for k in range(3):
        i.g[k] = float(g[k])
    o = Out()
    res = {}
    for name, nc in NCOMP.items():
        if (name in NEED_P and p is None) or (name in NEED_PHI and phi is None):
            continue
        res[name] = np.zeros((m.nFaces, nc) if nc > 1 else (m.nFaces,))
        setattr(o, name, res[name].ctypes.data_as(L.c_double_p))
    if call is None:
        L.check(L.lib.qgd_qhd_fluxes(dev._h, STENCIL_IDS[scheme], C.byref(i), C.byref(o)), "qgd_qhd_fluxes")
    else:
        call(scheme, i, o)
    return res


# ---- QGDCoeffs closures used by the QHD solvers (one line each on top of hQGD and qgdInterpolate) ---------------------
def tauQGDf(dev, model, aQGD=0.5, **par):
    """thermo.tauQGDf() for the QHD closures:
    constTau  tau = Tau                         constTau.C L71-74
    HbyUQHD   tau = aQGD*hQGD/UQHD              HbyUQHD.C L80-83
    T0byGr    tau = T0/Gr                       T0byGr.C L84-87
    H2bynuQHD tau = aQGD*hQGD^2/nu, nu = mu/rho H2bynuQHD.C L78-82   (pass nu=(cells, patch values))
    followed by tauQGDf = linearInterpolate(tauQGD)."""
    from . import fvsc

    m = dev.mesh
    h = fvsc.device_field(dev, "hQGD")
    hb = fvsc.device_field(dev, "hQGD.boundary")
    if model == "constTau":
        tau, taub = np.full(m.nCells, float(par["Tau"])), np.full(m.nBoundaryFaces, float(par["Tau"]))
    elif model == "HbyUQHD":
        tau, taub = aQGD * h / float(par["UQHD"]), aQGD * hb / float(par["UQHD"])
    elif model == "T0byGr":
        tau, taub = np.full(m.nCells, par["T0"] / par["Gr"]), np.full(m.nBoundaryFaces, par["T0"] / par["Gr"])
    elif model == "H2bynuQHD":
        nu, nub = par["nu"]
        tau, taub = aQGD * h * h / nu, aQGD * hb * hb / nub
    else:
        raise KeyError(model)
    return fvsc.qgdInterpolate(dev, fvsc.volField("tauQGD", tau, taub))
