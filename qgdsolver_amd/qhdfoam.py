"""Host-side mirror of the QHDFoam flux-assembly fragments over the C-ABI (qgd_qhd_fluxes).

``updateFluxes(...)`` covers QHDFoam/updateFields.H L36-73 + updateFluxes.H L33-38 (call it without p/phi before the
pressure equation) and the flux parts of QHDUEqn.H L36-43 / QHDTEqn.H L65-66 (call it again with p and phi).
``pEqn(...)`` is QHDpEqn.H L35-47 (qgd_qhd_pressure): the pressure Poisson equation assembled and solved on the device.
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


QhdInputs, QhdOutputs = _In, _Out   # qgd_qhd_inputs / qgd_qhd_outputs (host pointers for qgd_qhd_fluxes, device pointers for _dev)

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


class PoissonControl(C.Structure):
    """qgd_poisson_control: the fvSolution entries of p (tolerance, relTol, maxIter) + the reference level"""
    _fields_ = [("tolerance", C.c_double), ("relTol", C.c_double), ("maxIter", C.c_int32), ("pRefCell", C.c_int32),
                ("pRefValue", C.c_double)]


_P_KINDS = {"zeroGradient": L.BC_ZEROGRADIENT, "fixedValue": L.BC_FIXEDVALUE, "fixedGradient": L.BC_QGDFLUX, "qhdFlux": L.BC_QGDFLUX,
            "none": L.BC_NONE}


def qhdFluxGradient(phiwStar_b, tauQGDf_b, rhof_b, magSf_b):
    """gradient() of the qhdFlux patch field: -(phiwStar/tauQGDf*rhof/|Sf|) [qhdFluxFvPatchScalarField.C L193-203].
    (QHDFoam itself registers that flux as "phiwo", not "phiwStar", so there the lookup at L166-168 fails and the patch
    keeps the gradient read from its file; solvers that register phiwStar get this value.)"""
    return -(np.asarray(phiwStar_b) / np.asarray(tauQGDf_b) * np.asarray(rhof_b) / np.asarray(magSf_b))


def pEqn(dev, phiu, phiwo, taubyrhof, p, patch_kinds, pb=None, gradb=None, tolerance=1e-6, relTol=0.0, maxIter=1000,
         pRefCell=0, pRefValue=0.0, call=None):
    """QHDpEqn.H L35-47.  ``p``: initial guess (nCells); ``patch_kinds``: one word per patch (zeroGradient | fixedValue |
    fixedGradient/qhdFlux | none); ``pb`` / ``gradb``: patch values / patch-normal gradients (nBoundaryFaces).
    Returns (p, phi, info) with info = dict(iterations, initialResidual, finalResidual)."""
    m = dev.mesh

    def arr(a, n):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == n
        return a

    phiu, phiwo, tbr = arr(phiu, m.nFaces), arr(phiwo, m.nFaces), arr(taubyrhof, m.nFaces)
    pb, gradb = arr(pb, m.nBoundaryFaces), arr(gradb, m.nBoundaryFaces)
    kinds = np.asarray([_P_KINDS[k] for k in patch_kinds], dtype=np.int32)
    assert kinds.size == m.nPatches
    ctl = PoissonControl(float(tolerance), float(relTol), int(maxIter), int(pRefCell), float(pRefValue))
    p_out = np.array(p, dtype=np.float64, copy=True).reshape(-1)
    assert p_out.size == m.nCells
    phi = np.zeros(m.nFaces)
    info = np.zeros(3)
    dp = lambda a: a.ctypes.data_as(L.c_double_p) if a is not None and a.size else None  # noqa: E731
    if call is None:
        L.check(L.lib.qgd_qhd_pressure(dev._h, dp(phiu), dp(phiwo), dp(tbr), kinds.ctypes.data_as(L.c_int32_p), dp(pb), dp(gradb),
                                       C.byref(ctl), dp(p_out), dp(phi), dp(info)), "qgd_qhd_pressure")
    else:
        call(phiu, phiwo, tbr, kinds, pb, gradb, ctl, p_out, phi, info)
    return p_out, phi, dict(iterations=int(info[0]), initialResidual=float(info[1]), finalResidual=float(info[2]))


# ---- QHDFoam case resident on the device ------------------------------------------------------------------------------
TAU_MODELS = {"constTau": 0, "HbyUQHD": 1, "T0byGr": 2, "H2bynuQHD": 3}
_BC = {"zeroGradient": L.BC_ZEROGRADIENT, "fixedValue": L.BC_FIXEDVALUE, "slip": L.BC_SLIP, "fixedGradient": L.BC_QGDFLUX,
       "qhdFlux": L.BC_QGDFLUX, "qhdFluxCoupled": L.BC_QHDFLUX, "none": L.BC_NONE}


def qhd_options(**kw):
    """qgd_qhd_options with the library defaults; stencil and tauModel may be given as words, g as a 3-tuple"""
    o = L.QhdOptions()
    L.check(L.lib.qgd_qhd_options_default(C.byref(o)), "qgd_qhd_options_default")
    for k, v in kw.items():
        if k == "stencil" and isinstance(v, str):
            v = STENCIL_IDS[v]
        if k == "tauModel" and isinstance(v, str):
            v = TAU_MODELS[v]
        if k == "g":
            for i in range(3):
                o.g[i] = float(v[i])
            continue
        setattr(o, k, v)
    return o


class QHDFoamCase:
    """createFields.H + the while-loop body of QHDFoam.C L83-139 (both branches of implicitDiffusion) over the C-ABI (qgd_qhd_case_*)"""

    def __init__(self, dev, options=None):
        self.dev, self.mesh = dev, dev.mesh
        self.options = options if options is not None else qhd_options()
        h = C.c_void_p()
        L.check(L.lib.qgd_qhd_case_create(dev._h, C.byref(self.options), C.byref(h)), "qgd_qhd_case_create")
        self._handle = L.NativeHandle(h, L.lib.qgd_qhd_case_free)
        dev.adopt(self._handle)

    def set_bc(self, patch, U=("zeroGradient", None), T=("zeroGradient", None), p=("zeroGradient", None)):
        vu = np.asarray(U[1] if U[1] is not None else (0.0, 0.0, 0.0), dtype=np.float64)
        L.check(L.lib.qgd_qhd_case_set_bc(self._h, int(patch), _BC[U[0]], vu.ctypes.data_as(L.c_double_p), _BC[T[0]], float(T[1] or 0.0),
                                          _BC[p[0]], float(p[1] or 0.0)), "qgd_qhd_case_set_bc")

    def set_fields(self, U, T, p):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (U, T, p)]
        assert a[0].size == 3 * self.mesh.nCells and a[1].size == self.mesh.nCells and a[2].size == self.mesh.nCells
        L.check(L.lib.qgd_qhd_case_set_fields(self._h, *[x.ctypes.data_as(L.c_double_p) for x in a]), "qgd_qhd_case_set_fields")

    def step(self, n=1):
        L.check(L.lib.qgd_qhd_case_step(self._h, int(n)), "qgd_qhd_case_step")

    def field(self, name):
        base = name[:-len(".boundary")] if name.endswith(".boundary") else name
        nc = 3 if base == "U" else 1
        n = self.mesh.nBoundaryFaces if name.endswith(".boundary") else (self.mesh.nFaces if base in ("phi", "phiu", "phiwo", "tauQGDf")
                                                                         else self.mesh.nCells)
        out = np.zeros((n, nc) if nc > 1 else (n,))
        if n:
            L.check(L.lib.qgd_qhd_case_get_field(self._h, name.encode(), out.ctypes.data_as(L.c_double_p), out.size), f"qgd_qhd_case_get_field({name})")
        return out

    # ---- the step as phases (cell-range shards; the protocol is in include/qgd_amd.h and halo.QhdStepper) ---------------
    def step_phase(self, phase):
        L.check(L.lib.qgd_qhd_case_step_phase(self._h, int(phase)), "qgd_qhd_case_step_phase")

    def pending(self):
        """(action, device pointer, count) the phase in flight waits for before step_phase(9): 0 nothing, 1 halo message kind 3,
        2 / 3 SUM / MAX all-reduce of `count` doubles at the pointer (the multigrid hierarchy that spans the ranks, include/qgd_amd.h)"""
        a, p, n = C.c_int32(), C.c_void_p(), C.c_int64()
        L.check(L.lib.qgd_qhd_case_pending(self._h, C.byref(a), C.byref(p), C.byref(n)), "qgd_qhd_case_pending")
        return a.value, p.value, n.value

    def control_ptr(self):
        """device pointer of the 16-double control block of the pressure solve (slots [0,3) [3] [4] [5] [6,8) [8] are reduced)"""
        p = C.c_void_p()
        L.check(L.lib.qgd_qhd_case_control_ptr(self._h, C.byref(p)), "qgd_qhd_case_control_ptr")
        return p.value

    def control(self):
        """host copy of the control block (waits for the stream)"""
        a = np.zeros(16)
        L.check(L.lib.qgd_qhd_case_control(self._h, a.ctypes.data_as(L.c_double_p), 0), "qgd_qhd_case_control")
        return a

    def set_control(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == 16
        L.check(L.lib.qgd_qhd_case_control(self._h, a.ctypes.data_as(L.c_double_p), 1), "qgd_qhd_case_control")

    # ---- implicitDiffusion: the solve of the four systems {Ux, Uy, Uz, T} (its own 68-double control block) ----
    @property
    def implicit(self):
        return bool(self.options.implicitDiffusion)

    def implicit_control(self):
        a = np.zeros(68)
        L.check(L.lib.qgd_qhd_case_implicit_control(self._h, a.ctypes.data_as(L.c_double_p), 0), "qgd_qhd_case_implicit_control")
        return a

    def set_implicit_control(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == 68
        L.check(L.lib.qgd_qhd_case_implicit_control(self._h, a.ctypes.data_as(L.c_double_p), 1), "qgd_qhd_case_implicit_control")

    def implicit_control_ptr(self):
        p = C.c_void_p()
        L.check(L.lib.qgd_qhd_case_implicit_control_ptr(self._h, C.byref(p)), "qgd_qhd_case_implicit_control_ptr")
        return p.value

    def implicit_solve_done(self):
        a = (C.c_double * 2)()
        L.check(L.lib.qgd_qhd_case_implicit_solve_status(self._h, a), "qgd_qhd_case_implicit_solve_status")
        return a[0] != 0.0

    def implicit_info(self):
        """iterations / initial / final residual of Ux, Uy, Uz, T in the last step (what OpenFOAM prints per solve), the steps in which
        a solve stopped above implicitTol, and the algorithm (QGD_IMPL_SOLVER)"""
        a = (C.c_double * 16)()
        L.check(L.lib.qgd_qhd_case_implicit_info(self._h, a), "qgd_qhd_case_implicit_info")
        names = ("Ux", "Uy", "Uz", "T")
        return dict(implicit=a[13] != 0.0, solver={0: None, 1: "pcg", 2: "chebyshev"}[int(a[13])], unconverged_steps=int(a[12]), stalled_steps=int(a[14]),
                    solves={n: dict(iterations=int(a[k]), initial=a[4 + k], final=a[8 + k]) for k, n in enumerate(names)})

    def solve_status(self):
        a = (C.c_double * 4)()
        L.check(L.lib.qgd_qhd_case_solve_status(self._h, a), "qgd_qhd_case_solve_status")
        return dict(done=int(a[0]), iterations=int(a[1]), initialResidual=a[2], finalResidual=a[3])

    def sync(self):
        L.check(L.lib.qgd_qhd_case_sync(self._h), "qgd_qhd_case_sync")

    def sweep_time(self, reps=20):
        """average ms of one level-0 smoothing sweep of the pressure multigrid (HIP events), its rows, ELL width, bytes per value"""
        a = (C.c_double * 4)()
        L.check(L.lib.qgd_qhd_case_sweep_time(self._h, int(reps), a), "qgd_qhd_case_sweep_time")
        return dict(ms=a[0], rows=int(a[1]), width=float(a[2]), value_bytes=int(a[3]))

    def halo_count(self, slot, kind):
        s, r = C.c_int64(), C.c_int64()
        L.check(L.lib.qgd_qhd_case_halo_count(self._h, int(slot), int(kind), C.byref(s), C.byref(r)), "qgd_qhd_case_halo_count")
        return s.value, r.value

    def halo_buffer(self, n):
        """device buffer of n doubles (released with the device)"""
        return self.dev.alloc(8 * max(int(n), 1))

    def halo_pack(self, slot, kind, dev_ptr):
        L.check(L.lib.qgd_qhd_case_halo_pack(self._h, int(slot), int(kind), C.c_void_p(dev_ptr)), "qgd_qhd_case_halo_pack")

    def halo_unpack(self, slot, kind, dev_ptr):
        L.check(L.lib.qgd_qhd_case_halo_unpack(self._h, int(slot), int(kind), C.c_void_p(dev_ptr)), "qgd_qhd_case_halo_unpack")

    def info(self):
        a = (C.c_double * 8)()
        L.check(L.lib.qgd_qhd_case_info(self._h, a), "qgd_qhd_case_info")
        return dict(time=a[0], deltaT=a[1], pIterations=int(a[2]), pInitialResidual=a[3], pFinalResidual=a[4], steps=int(a[5]),
                    mgLevels=int(a[6]), pSolveMs=a[7])

    def fused_info(self):
        """which parts of the step run on the cell blocks of QGDFoam's one-launch step (qgd_qhd_case_fused_info)"""
        a = (C.c_int64 * 4)()
        L.check(L.lib.qgd_qhd_case_fused_info(self._h, a), "qgd_qhd_case_fused_info")
        return dict(fusedAdvance=bool(a[0] & 1), fusedAssemble=bool(a[0] & 2), blocks=int(a[1]), ldsAdvance=int(a[2]), ldsAssemble=int(a[3]))

    @property
    def _h(self):
        return self._handle.value

    def close(self):
        if getattr(self, "_handle", None):
            self._handle.free()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
