"""Host-side mirror of the QGDFoam loop pieces over the C-ABI.

``QGDFoamCase`` holds what createFields.H / createFaceFields.H / createFaceFluxes.H
create; ``updateFluxes()`` is updateFields.H + updateFluxes.H; ``step()`` is the
``while (runTime.run())`` body (QGDFoam.C L90-163).  ``thermo`` exposes the
QGDThermo accessor names (QGDThermo.H L99-135).
"""
import ctypes as C

import numpy as np

from . import _lib as L

_VECTOR_FIELDS = {"U": 3, "rhoU": 3, "phiJmU": 3, "phiP": 3, "phiPi": 3, "gradUf": 9, "gradef": 3, "gradRhof": 3, "gradPf": 3, "phiTauMC": 3}
_FACE_FIELDS = {"phiJm", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU", "phiwStar", "phi", "tauQGDf", "hQGDf",
                "gradUf", "gradef", "gradRhof", "gradPf", "phiTauMC", "phiSigmaDotU"}

STENCIL_IDS = {"reduced": L.FVSC_REDUCED, "leastSquares": L.FVSC_LEASTSQUARES, "leastSquaresOpt": L.FVSC_LEASTSQUARES,
               "GaussVolPoint": L.FVSC_GAUSSVOLPOINT}


TERM_ORDER = ("grad(U)", "grad(e)", "grad(rho)", "grad(p)")   # qgd_case_options::termStencil [QGDFoam/updateFluxes.H L41-65]


def default_options(**kw):
    """qgd_case_options with the library defaults.  ``stencil`` may be a word; ``termStencils`` a dict {"grad(p)": "reduced", ...} of
    per-term fvsc entries [fvsc.C L51-58] (terms not named take ``stencil``)"""
    o = L.CaseOptions()
    L.check(L.lib.qgd_case_options_default(C.byref(o)), "qgd_case_options_default")
    for k, v in kw.items():
        if k == "stencil" and isinstance(v, str):
            v = STENCIL_IDS[v]
        if k == "termStencils":
            for term, word in (v or {}).items():
                o.termStencil[TERM_ORDER.index(term)] = 1 + (STENCIL_IDS[word] if isinstance(word, str) else int(word))
            continue
        setattr(o, k, v)
    return o


class QGDThermo:
    """Accessor surface of QGDThermo (QGDThermo.H L99-135)."""

    def __init__(self, case):
        self._c = case

    def tauQGDf(self):
        return self._c.field("tauQGDf")

    def hQGDf(self):
        return self._c.field("hQGDf")

    def tauQGD(self):
        return self._c.field("tauQGD")

    def hQGD(self):
        return self._c.field("hQGD")

    def muQGD(self):
        return self._c.field("muQGD")

    def alphauQGD(self):
        return self._c.field("alphauQGD")

    def c(self):
        return self._c.field("c")

    def p(self):
        return self._c.field("p")

    def rho(self):
        return self._c.field("rho")

    def mu(self):
        return self._c.field("mu")

    def implicitDiffusion(self):
        return bool(self._c.options.implicitDiffusion)


class QGDFoamCase:
    def __init__(self, dev, options=None):
        self.dev = dev
        self.mesh = dev.mesh
        self.options = options if options is not None else default_options()
        h = C.c_void_p()
        L.check(L.lib.qgd_case_create(dev._h, C.byref(self.options), C.byref(h)), "qgd_case_create")
        self._handle = L.NativeHandle(h, L.lib.qgd_case_free)
        dev.adopt(self._handle)
        self.thermo = QGDThermo(self)

    def set_bc(self, patch, U=("zeroGradient", None), T=("zeroGradient", None), p=("zeroGradient", None)):
        kinds = {"zeroGradient": L.BC_ZEROGRADIENT, "fixedValue": L.BC_FIXEDVALUE, "slip": L.BC_SLIP, "qgdFlux": L.BC_QGDFLUX,
                 "none": L.BC_NONE}
        vu = np.asarray(U[1] if U[1] is not None else (0.0, 0.0, 0.0), dtype=np.float64)
        L.check(L.lib.qgd_case_set_bc(self._h, patch, kinds[U[0]], vu.ctypes.data_as(L.c_double_p), kinds[T[0]],
                                      float(T[1] or 0.0), kinds[p[0]], float(p[1] or 0.0)), "qgd_case_set_bc")

    def set_fields(self, U, T, p):
        U = np.ascontiguousarray(U, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert U.size == 3 * self.mesh.nCells and T.size == self.mesh.nCells and p.size == self.mesh.nCells
        L.check(L.lib.qgd_case_set_fields(self._h, U.ctypes.data_as(L.c_double_p), T.ctypes.data_as(L.c_double_p),
                                          p.ctypes.data_as(L.c_double_p)), "qgd_case_set_fields")

    def set_qgd_coeffs(self, alphaQGD=None, ScQGD=None):
        """non-uniform alphaQGD / ScQGD fields (QGDCoeffs.C L119-160, constScPrModel1.C L66-79): each a pair
        (cell values, patch values) or None for the uniform value of the options; call before set_fields"""
        keep = []

        def ptrs(pair):
            if pair is None:
                return None, None
            c = np.ascontiguousarray(pair[0], dtype=np.float64)
            b = np.ascontiguousarray(pair[1], dtype=np.float64)
            assert c.size == self.mesh.nCells and b.size == self.mesh.nBoundaryFaces
            if b.size == 0:
                b = np.zeros(1)
            keep.extend([c, b])
            return c.ctypes.data_as(L.c_double_p), b.ctypes.data_as(L.c_double_p)

        a, ab = ptrs(alphaQGD)
        s, sb = ptrs(ScQGD)
        L.check(L.lib.qgd_case_set_qgd_coeffs(self._h, a, ab, s, sb), "qgd_case_set_qgd_coeffs")

    def updateFluxes(self):
        L.check(L.lib.qgd_case_update_fluxes(self._h), "qgd_case_update_fluxes")

    def step(self, n=1):
        """n whole steps of an unsharded case; returns when the device has finished (qgd_case_step)"""
        L.check(L.lib.qgd_case_step(self._h, int(n)), "qgd_case_step")

    def step_phase(self, phase):
        """one stream-ordered phase of a step, no host synchronisation (qgd_case_step_phase): 0 assembly, 1 advance (10 + 11: the boundary
        layer of a shard first), 5 + 6 the assembly split around the mid-step exchange, 3 = one whole step of an unsharded case, 20..35 the
        implicitDiffusion branch; the exchanges between the phases are the caller's (halo.py)"""
        L.check(L.lib.qgd_case_step_phase(self._h, int(phase)), "qgd_case_step_phase")

    def reduction_ptr(self):
        """device pointer of the 2-double {max Cof, -min tauQGDf} buffer (valid between phase 0 and phase 1)"""
        p = C.c_void_p()
        L.check(L.lib.qgd_case_reduction_ptr(self._h, C.byref(p)), "qgd_case_reduction_ptr")
        return p.value

    def set_stream(self, raw_stream):
        """Run on a caller-owned hipStream_t (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        L.check(L.lib.qgd_case_set_stream(self._h, C.c_void_p(raw_stream)), "qgd_case_set_stream")

    def set_halo_stream(self, raw_stream):
        L.check(L.lib.qgd_case_set_halo_stream(self._h, C.c_void_p(raw_stream)), "qgd_case_set_halo_stream")

    def sync(self):
        L.check(L.lib.qgd_case_stream_sync(self._h), "qgd_case_stream_sync")

    def field(self, name):
        base = name[:-len(".boundary")] if name.endswith(".boundary") else name
        nc = _VECTOR_FIELDS.get(base, 1)
        if name.endswith(".boundary"):
            n = self.mesh.nBoundaryFaces
        elif base in _FACE_FIELDS:
            n = self.mesh.nFaces
        else:
            n = self.mesh.nCells
        out = np.zeros((n, nc) if nc > 1 else (n,), dtype=np.float64)
        if n:
            L.check(L.lib.qgd_case_get_field(self._h, name.encode(), out.ctypes.data_as(L.c_double_p), out.size),
                    f"qgd_case_get_field({name})")
        return out

    def info(self):
        a = (C.c_double * 6)()
        L.check(L.lib.qgd_case_info(self._h, a), "qgd_case_info")
        return dict(time=a[0], deltaT=a[1], CoNum=a[2], minRho=a[3], minE=a[4], steps=int(a[5]))

    def fused_info(self):
        """whether the case advances with the fused kernel of the explicit step (vertex values, faces and cell update of a block of cells in
        one launch), its blocks, faces computed / cell records staged / vertex values formed per step, LDS bytes (qgd_case_fused_info)"""
        a = (C.c_int64 * 8)()
        L.check(L.lib.qgd_case_fused_info(self._h, a), "qgd_case_fused_info")
        return dict(fused=int(a[0]) == 1, fusedImplicit=int(a[0]) == 2, fusedAdjust=int(a[0]) == 3, blocks=int(a[1]), facesComputed=int(a[2]), ldsBytes=int(a[3]), cellsStaged=int(a[4]),
                    cellsStagedFull=int(a[5]), verticesFormed=int(a[6]), layerBlocks=int(a[7]))

    def implicit_info(self):
        """the four linear solves of the implicitDiffusion branch in the last step (qgd_case_implicit_info)"""
        a = (C.c_double * 16)()
        L.check(L.lib.qgd_case_implicit_info(self._h, a), "qgd_case_implicit_info")
        names = ("Ux", "Uy", "Uz", "e")
        return dict(implicit=bool(a[13]), solver={0: None, 1: "pcg", 2: "chebyshev"}[int(a[13])], unconverged_steps=int(a[12]), stalled_steps=int(a[14]),
                    solves={n: dict(iterations=int(a[k]), initial=a[4 + k], final=a[8 + k]) for k, n in enumerate(names)})

    def implicit_apply_time(self, reps=20):
        """measurement: average ms over `reps` launches of the kernel the branch spends most of its time in -- one Chebyshev step of the
        U system (iChebKernel<3,0>) or, with QGD_IMPL_SOLVER=pcg, its matrix product (iApplyKernel<3,1>) -- and its rows"""
        a = (C.c_double * 2)()
        L.check(L.lib.qgd_case_implicit_apply_time(self._h, int(reps), a), "qgd_case_implicit_apply_time")
        return dict(ms=a[0], rows=int(a[1]))

    # ---- the implicitDiffusion branch on shards (phases 20..35 of qgd_case_step_phase; halo.ImplicitStepper drives them) ----
    def implicit_control(self):
        """host copy of the 68-double control block of the solve in flight (slot-major: [slot * 4 + component])"""
        a = np.zeros(68)
        L.check(L.lib.qgd_case_implicit_control(self._h, a.ctypes.data_as(L.c_double_p), 0), "qgd_case_implicit_control")
        return a

    def set_implicit_control(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == 68
        L.check(L.lib.qgd_case_implicit_control(self._h, a.ctypes.data_as(L.c_double_p), 1), "qgd_case_implicit_control")

    def implicit_control_ptr(self):
        p = C.c_void_p()
        L.check(L.lib.qgd_case_implicit_control_ptr(self._h, C.byref(p)), "qgd_case_implicit_control_ptr")
        return p.value

    def implicit_solve_done(self):
        a = (C.c_double * 2)()
        L.check(L.lib.qgd_case_implicit_solve_status(self._h, a), "qgd_case_implicit_solve_status")
        return a[0] != 0.0

    # ---- the message in the middle of the flux assembly (step phases 5 | 6 instead of 0; include/qgd_amd.h) ----
    def needs_mid_exchange(self):
        n = C.c_int32()
        L.check(L.lib.qgd_case_mid_exchange_needed(self._h, C.byref(n)), "qgd_case_mid_exchange_needed")
        return bool(n.value)

    def mid_halo_count(self, slot):
        s, r = C.c_int64(), C.c_int64()
        L.check(L.lib.qgd_case_mid_halo_count(self._h, int(slot), C.byref(s), C.byref(r)), "qgd_case_mid_halo_count")
        return s.value, r.value

    def mid_halo_pack(self, slot, dev_ptr):
        L.check(L.lib.qgd_case_mid_halo_pack(self._h, int(slot), C.c_void_p(dev_ptr)), "qgd_case_mid_halo_pack")

    def mid_halo_unpack(self, slot, dev_ptr):
        L.check(L.lib.qgd_case_mid_halo_unpack(self._h, int(slot), C.c_void_p(dev_ptr)), "qgd_case_mid_halo_unpack")

    def implicit_halo_count(self, slot, kind):
        s, r = C.c_int64(), C.c_int64()
        L.check(L.lib.qgd_case_implicit_halo_count(self._h, int(slot), int(kind), C.byref(s), C.byref(r)), "qgd_case_implicit_halo_count")
        return s.value, r.value

    def implicit_halo_pack(self, slot, kind, dev_ptr):
        L.check(L.lib.qgd_case_implicit_halo_pack(self._h, int(slot), int(kind), C.c_void_p(dev_ptr)), "qgd_case_implicit_halo_pack")

    def implicit_halo_unpack(self, slot, kind, dev_ptr):
        L.check(L.lib.qgd_case_implicit_halo_unpack(self._h, int(slot), int(kind), C.c_void_p(dev_ptr)), "qgd_case_implicit_halo_unpack")

    def halo_buffer(self, n):
        """device buffer of n doubles (released with the device)"""
        return self.dev.alloc(8 * max(int(n), 1))

    # ---- halo ------------------------------------------------------------------
    def halo_count(self, slot):
        """doubles in the message sent to the neighbour behind halo slot ``slot``"""
        n = C.c_int64()
        L.check(L.lib.qgd_case_halo_count(self._h, slot, C.byref(n)), "qgd_case_halo_count")
        return n.value

    def halo_recv_count(self, slot):
        n = C.c_int64()
        L.check(L.lib.qgd_case_halo_recv_count(self._h, slot, C.byref(n)), "qgd_case_halo_recv_count")
        return n.value

    def halo_pack(self, side, dev_ptr):
        L.check(L.lib.qgd_case_halo_pack(self._h, side, C.c_void_p(dev_ptr)), "qgd_case_halo_pack")

    def halo_unpack(self, side, dev_ptr):
        L.check(L.lib.qgd_case_halo_unpack(self._h, side, C.c_void_p(dev_ptr)), "qgd_case_halo_unpack")

    # ---- measurement --------------------------------------------------------------
    def timing(self, enable=True):
        L.check(L.lib.qgd_case_timing(self._h, 1 if enable else 0), "qgd_case_timing")

    def timing_reset(self):
        L.check(L.lib.qgd_case_timing_reset(self._h), "qgd_case_timing_reset")

    def kernel_time(self, k):
        ms = C.c_double()
        n = C.c_int64()
        L.check(L.lib.qgd_case_kernel_time(self._h, k, C.byref(ms), C.byref(n)), "qgd_case_kernel_time")
        return ms.value, n.value

    def device_bytes(self):
        n = C.c_int64()
        L.check(L.lib.qgd_case_device_bytes(self._h, C.byref(n)), "qgd_case_device_bytes")
        return n.value

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


def speciesFlux(dev, scheme, Y, U, phiJm, phi, tauQGDf, call=None):
    """Species block of reactingLagrangianQGDFoam/updateFluxes.H L117-132 for one species.
    Y, U: (internal, boundary) pairs; returns dict(phiJmY, diffusiveFlux, gradYf)."""
    m = dev.mesh
    a = lambda x, n: np.ascontiguousarray(x, dtype=np.float64).reshape(-1) if n else np.zeros(1)  # noqa: E731
    Yc, Yb, Uc, Ub = a(Y[0], m.nCells), a(Y[1], m.nBoundaryFaces), a(U[0], m.nCells), a(U[1], m.nBoundaryFaces)
    jm, ph, tau = a(phiJm, 1), a(phi, 1), a(tauQGDf, 1)
    assert Yc.size == m.nCells and Uc.size == 3 * m.nCells and jm.size == m.nFaces and ph.size == m.nFaces and tau.size == m.nFaces
    out = dict(phiJmY=np.zeros(m.nFaces), diffusiveFlux=np.zeros(m.nFaces), gradYf=np.zeros((m.nFaces, 3)))
    dp = lambda x: x.ctypes.data_as(L.c_double_p)  # noqa: E731
    if call is None:
        sid = STENCIL_IDS[scheme] if isinstance(scheme, str) else int(scheme)
        L.check(L.lib.qgd_species_flux(dev._h, sid, dp(Yc), dp(Yb), dp(Uc), dp(Ub), dp(jm), dp(ph), dp(tau), dp(out["phiJmY"]),
                                       dp(out["diffusiveFlux"]), dp(out["gradYf"])), "qgd_species_flux")
    else:
        call(scheme, Yc, Yb, Uc, Ub, jm, ph, tau, out["phiJmY"], out["diffusiveFlux"], out["gradYf"])
    return out


def speciesStep(dev, Y, rhoOld, rho, phiJmY, muf, Sc, deltaT, diffusiveFlux, Su=None, call=None):
    """One species of QGDYEqn.H L67-86 (explicit branch; the sources as one explicit field Su or None): returns the new cell values of
    Yi (already clipped at 0) and adds (muf/Sc) snGrad(Yi.old) |Sf| to diffusiveFlux in place.  Y: (internal, boundary) pair."""
    m = dev.mesh
    a = lambda x, n: np.ascontiguousarray(x, dtype=np.float64).reshape(-1) if n else np.zeros(1)  # noqa: E731
    Yc, Yb = a(Y[0], m.nCells), a(Y[1], m.nBoundaryFaces)
    ro, rn, jm, mf = a(rhoOld, 1), a(rho, 1), a(phiJmY, 1), a(muf, 1)
    su = None if Su is None else a(Su, 1)
    assert Yc.size == m.nCells and ro.size == m.nCells and rn.size == m.nCells and jm.size == m.nFaces and mf.size == m.nFaces
    assert isinstance(diffusiveFlux, np.ndarray) and diffusiveFlux.dtype == np.float64 and diffusiveFlux.size == m.nFaces and diffusiveFlux.flags.c_contiguous
    Ynew = np.zeros(m.nCells)
    dp = lambda x: x.ctypes.data_as(L.c_double_p)  # noqa: E731
    if call is None:
        L.check(L.lib.qgd_species_step(dev._h, dp(Yc), dp(Yb), dp(ro), dp(rn), dp(jm), dp(mf), float(Sc), float(deltaT),
                                       dp(su) if su is not None else None, dp(diffusiveFlux), dp(Ynew)), "qgd_species_step")
    else:
        call(Yc, Yb, ro, rn, jm, mf, float(Sc), float(deltaT), su, diffusiveFlux, Ynew)
    return Ynew


def speciesStepImplicit(dev, Y, rhoOld, rho, phiJmY, muf, Sc, deltaT, diffusiveFlux, Su=None, fixedValueFaces=None, tolerance=1e-10, maxIter=1000,
                        call=None):
    """One species of QGDYEqn.H L47-66 (the implicitDiffusion branch: fvm::laplacian(muf/Sc, Yi), diffusiveFlux += YEqn.flux()): returns
    (new cell values of Yi clipped at 0, {iterations, initial, final}).  fixedValueFaces: boolean mask over the boundary faces of the
    fixedValue patches of Yi (their values are Y[1]); all other patch faces are zeroGradient."""
    m = dev.mesh
    a = lambda x, n: np.ascontiguousarray(x, dtype=np.float64).reshape(-1) if n else np.zeros(1)  # noqa: E731
    Yc, Yb = a(Y[0], m.nCells), a(Y[1], m.nBoundaryFaces)
    ro, rn, jm, mf = a(rhoOld, 1), a(rho, 1), a(phiJmY, 1), a(muf, 1)
    su = None if Su is None else a(Su, 1)
    fx = None if fixedValueFaces is None else np.ascontiguousarray(fixedValueFaces, dtype=np.uint8).reshape(-1)
    assert Yc.size == m.nCells and ro.size == m.nCells and rn.size == m.nCells and jm.size == m.nFaces and mf.size == m.nFaces
    assert fx is None or fx.size == m.nBoundaryFaces
    assert isinstance(diffusiveFlux, np.ndarray) and diffusiveFlux.dtype == np.float64 and diffusiveFlux.size == m.nFaces and diffusiveFlux.flags.c_contiguous
    Ynew, info = np.zeros(m.nCells), np.zeros(3)
    dp = lambda x: x.ctypes.data_as(L.c_double_p)  # noqa: E731
    if call is None:
        L.check(L.lib.qgd_species_step_implicit(dev._h, dp(Yc), dp(Yb), fx.ctypes.data_as(C.c_void_p) if fx is not None and fx.size else None, dp(ro),
                                                dp(rn), dp(jm), dp(mf), float(Sc), float(deltaT), dp(su) if su is not None else None,
                                                float(tolerance), int(maxIter), dp(diffusiveFlux), dp(Ynew), dp(info)), "qgd_species_step_implicit")
    else:
        call(Yc, Yb, fx, ro, rn, jm, mf, float(Sc), float(deltaT), su, float(tolerance), int(maxIter), diffusiveFlux, Ynew, info)
    return Ynew, dict(iterations=int(info[0]), initial=float(info[1]), final=float(info[2]))


def QGDYEqn(dev, Y, rhoOld, rho, phiJmY, muf, ScNumbers, deltaT, diffusiveFlux, inertIndex, active=None, Su=None, call=None, implicitDiffusion=False,
            fixedValueFaces=None, tolerance=1e-10, maxIter=1000):
    """QGDYEqn.H L38-92 over all species (implicitDiffusion: L47-66 through speciesStepImplicit, fixedValueFaces[i] = mask of species i's
    fixedValue patch faces; else the explicit branch L67-86): Y[i] = (internal, boundary) pairs (old time level), phiJmY[i], diffusiveFlux[i]
    per species (numpy arrays over the faces, updated in place), Su[i] explicit sources or None.  Returns the list of new cell fields:
    the active species from speciesStep, the inert one as 1 - sum of the others, clipped at 0 [L86-91]."""
    n = len(Y)
    new = [None] * n
    Yt = np.zeros(dev.mesh.nCells)                                # volScalarField Yt(0.0*Y[0])
    for i in range(n):
        if i != inertIndex and (active is None or active[i]):
            if implicitDiffusion:
                new[i], _ = speciesStepImplicit(dev, Y[i], rhoOld, rho, phiJmY[i], muf, ScNumbers[i], deltaT, diffusiveFlux[i], None if Su is None else Su[i],
                                                None if fixedValueFaces is None else fixedValueFaces[i], tolerance, maxIter, call)
            else:
                new[i] = speciesStep(dev, Y[i], rhoOld, rho, phiJmY[i], muf, ScNumbers[i], deltaT, diffusiveFlux[i], None if Su is None else Su[i], call)
            diffusiveFlux[inertIndex] -= diffusiveFlux[i]           # L65 / L83 (as listed: the running total of species i, not this step's increment)
            Yt += new[i]
    for i in range(n):
        if new[i] is None and i != inertIndex:
            new[i] = np.array(Y[i][0], dtype=float)                # inactive species keep their values
    new[inertIndex] = np.maximum(1.0 - Yt, 0.0)                    # L90-91
    return new
