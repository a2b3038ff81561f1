"""Host-side mirror of the reference's ``fvsc`` namespace over the C-ABI.

Same names and argument meaning as the reference so that tests read like the
reference's own call sites:

* ``fvsc.grad(vf)`` / ``fvsc.div(vf)``           -- fvsc.H L46-68, fvsc.C L87-167
* ``fvscStencil.New`` / ``fvscStencil.lookupOrNew`` with a run-time selection
  table keyed by the scheme word                 -- fvscStencil.C L59-118
* scheme word resolution from ``fvSchemes['fvsc']`` (per-term entry, else
  ``default``) and its checks                    -- fvsc.C L47-85

(listing lines of /root/reference/docs/html/<file>_source.html)
"""
import ctypes as C

import numpy as np

from . import _lib as L


class Device:
    """qgd_device_t: the mesh uploaded to one GPU (plays the role of fvMesh + objectRegistry)."""

    def __init__(self, mesh, device_id=0, fv_schemes=None, fused_tables=True):
        """fused_tables=False: do not build the block tables of QGDFoam's fused explicit step (seconds of set-up and GBs of device memory a
        QHDFoam case or an implicitDiffusion case never uses); "any": build and use them whatever the blocks look like (tests, probes: the
        default leaves meshes whose blocks come out small to the separate kernels); True: the library's default, which the QGD_FUSED
        environment variable can still override (qgd_device_create_with)"""
        self.mesh = mesh
        self.device_id = device_id
        h = C.c_void_p()
        flags = {True: 0, False: L.DEVICE_NO_FUSED_TABLES, "any": L.DEVICE_FUSED_ANY_BLOCKS}[fused_tables]
        L.check(L.lib.qgd_device_create_with(mesh._h, device_id, flags, C.byref(h)), "qgd_device_create_with")
        self._h = h
        self._case_handles = []   # native handles of the cases created on this device: close() frees them first (see adopt)
        self.fvSchemes = fv_schemes if fv_schemes is not None else {"fvsc": {"default": "GaussVolPoint"}}
        self._registry = {}  # objectRegistry of stencils by name

    def alloc(self, nbytes):
        """raw device buffer (zeroed); returns the device pointer as int"""
        p = C.c_void_p()
        L.check(L.lib.qgd_device_alloc(self._h, int(nbytes), C.byref(p)), "qgd_device_alloc")
        return p.value

    def release(self, ptr):
        L.check(L.lib.qgd_device_release(self._h, C.c_void_p(ptr)), "qgd_device_release")

    def sync(self):
        """wait for everything queued on the device handle's stream (the *_dev operator entries are stream-ordered)"""
        L.check(L.lib.qgd_device_sync(self._h), "qgd_device_sync")

    def to_device(self, array):
        """copy a host array into a fresh device buffer; returns the device pointer"""
        a = np.ascontiguousarray(array, dtype=np.float64)
        ptr = self.alloc(max(a.nbytes, 8))
        if a.nbytes:
            L.check(L.lib.qgd_device_copy(self._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "qgd_device_copy")
        return ptr

    def to_host(self, ptr, shape):
        out = np.zeros(shape, dtype=np.float64)
        if out.nbytes:
            L.check(L.lib.qgd_device_copy(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 0), "qgd_device_copy")
        return out

    def lsq_stencil(self, face, cap=64):
        """cells of the leastSquares stencil of an internal face in the order the device sums them (qgd_device_lsq_stencil)"""
        out = (C.c_int32 * cap)()
        n = C.c_int32()
        L.check(L.lib.qgd_device_lsq_stencil(self._h, int(face), out, cap, C.byref(n)), "qgd_device_lsq_stencil")
        return [int(out[i]) for i in range(min(n.value, cap))]

    def fused_blocks(self):
        """the block tables of the fused explicit step on this device (qgd_device_fused_blocks)"""
        a = (C.c_int64 * 8)()
        L.check(L.lib.qgd_device_fused_blocks(self._h, a), "qgd_device_fused_blocks")
        b = int(a[7])
        return dict(blocks=int(a[0]), layerBlocks=int(a[1]), templates=int(a[2]), ldsBytes=int(a[3]), blockListBytes=int(a[4]),
                    templateBytes=int(a[5]), buildSeconds=a[6] / 1e3, brick=(b & 255, (b >> 8) & 255, (b >> 16) & 255))

    def face_tiles(self):
        """how the internal faces go through the 3-D GaussVolPoint flux kernel (qgd_device_face_tiles)"""
        a = (C.c_int64 * 4)()
        L.check(L.lib.qgd_device_face_tiles(self._h, a), "qgd_device_face_tiles")
        return dict(facesPerTile=a[0], tiles=a[1], gatherTiles=a[2], ldsBytes=a[3])

    def adopt(self, handle):
        """A case keeps a pointer to its device inside the library, so it has to be freed BEFORE the device.  A reference from the
        case to the device orders that under reference counting, but not when both die in one pass of the cycle collector, which
        runs finalisers in any order and clears weak references first (seen: qgd_case_free reading the freed device, `invalid
        device ordinal`).  The device therefore holds the cases' native handles (NativeHandle: freed once, by whoever comes first)
        and frees those still open before it frees itself."""
        self._case_handles = [h for h in self._case_handles if h.value] + [handle]

    def close(self):
        if getattr(self, "_h", None):
            for h in getattr(self, "_case_handles", ()):
                h.free()
            self._case_handles = []
            # qgd_device_free refuses (and frees nothing) while a case created on the device is still open -- one made through the
            # raw C entry and never adopted here, say.  Keep the handle then, so the free can be retried, and say so.
            rc = L.lib.qgd_device_free(self._h)
            if rc != 0:
                raise RuntimeError("qgd_device_free: " + L.lib.qgd_last_error().decode())
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class volField:
    """Minimal vol<Type>Field: name, internal (nCells x ncomp) and boundary (nBoundaryFaces x ncomp) values."""

    def __init__(self, name, internal, boundary):
        self.name = name
        self.internal = np.ascontiguousarray(internal, dtype=np.float64)
        self.boundary = np.ascontiguousarray(boundary, dtype=np.float64)
        self.ncomp = 1 if self.internal.ndim == 1 else self.internal.shape[1]


class deviceVolField:
    """A vol<Type>Field whose internal and patch values already live in device memory (pointers of ``Device.alloc`` /
    ``Device.to_device``): the operators take it through the ``*_dev`` entries -- nothing is staged, nothing crosses PCIe --
    and return the DEVICE pointer of the face field they wrote into ``out`` (nFaces x ncompOut doubles, caller-owned)."""

    def __init__(self, name, internal_ptr, boundary_ptr, ncomp):
        self.name, self.internal, self.boundary, self.ncomp = name, int(internal_ptr), int(boundary_ptr), int(ncomp)


class fvscStencil:
    """Plugin base + run-time selection table ``components`` (fvscStencil.H L46-137)."""

    componentsConstructorTable = {}
    typeName = "fvscStencil"

    def __init__(self, dev, word):
        self.dev = dev
        self.word = word
        sid = C.c_int()
        L.check(L.lib.qgd_stencil_lookup(dev._h, word.encode(), C.byref(sid)), f"fvscStencil::New({word})")
        self.stencil_id = sid.value

    @classmethod
    def New(cls, word, dev):
        ctor = cls.componentsConstructorTable.get(word)
        if ctor is None:
            # same failure text shape as fvscStencil.C L72-78; goes through the C-ABI for the status code
            sid = C.c_int()
            L.check(L.lib.qgd_stencil_lookup(dev._h, word.encode(), C.byref(sid)), f"fvscStencil::New({word})")
            raise L.QgdError(L.ERR_UNKNOWN_NAME, f"fvscStencil::New({word})")
        return ctor(dev, word)

    @classmethod
    def lookupOrNew(cls, word, dev):
        if word not in dev._registry:
            dev._registry[word] = cls.New(word, dev)
        return dev._registry[word]

    def _op(self, fn, vf, ncomp_in, ncomp_out, out=None):
        m = self.dev.mesh
        if vf.ncomp != ncomp_in:
            raise ValueError(f"{fn}: field has {vf.ncomp} components, expected {ncomp_in}")
        if isinstance(vf, deviceVolField):
            if out is None:
                raise ValueError(f"{fn}_dev: a device-resident field needs the device pointer of its result (out=)")
            L.check(getattr(L.lib, fn + "_dev")(self.dev._h, self.stencil_id, C.c_void_p(vf.internal), C.c_void_p(vf.boundary),
                                                C.c_void_p(out)), fn + "_dev")
            return out
        out = np.zeros((m.nFaces, ncomp_out) if ncomp_out > 1 else (m.nFaces,), dtype=np.float64)
        bnd = vf.boundary if vf.boundary.size else np.zeros(1)
        L.check(getattr(L.lib, fn)(self.dev._h, self.stencil_id, vf.internal.ctypes.data_as(L.c_double_p),
                                   bnd.ctypes.data_as(L.c_double_p), out.ctypes.data_as(L.c_double_p)), fn)
        return out

    # the four virtuals (fvscStencil.H L105-130)
    def Grad(self, vf, out=None):
        if vf.ncomp == 1:
            return self._op("qgd_fvsc_grad_s", vf, 1, 3, out)
        if vf.ncomp == 3:
            return self._op("qgd_fvsc_grad_v", vf, 3, 9, out)
        raise L.QgdError(L.ERR_NOT_IMPLEMENTED, "Grad of a field that is neither scalar nor vector")

    def Div(self, vf, out=None):
        if vf.ncomp == 3:
            return self._op("qgd_fvsc_div_v", vf, 3, 1, out)
        if vf.ncomp == 9:
            return self._op("qgd_fvsc_div_t", vf, 9, 3, out)
        raise L.QgdError(L.ERR_NOT_IMPLEMENTED, "Div of a field that is neither vector nor tensor")


for _w in ("reduced", "leastSquares", "leastSquaresOpt", "GaussVolPoint"):
    fvscStencil.componentsConstructorTable[_w] = fvscStencil  # addToRunTimeSelectionTable(fvscStencil, <T>, components)


def fvscOpName(dev, term_name):
    """fvsc.C L47-85: per-term entry of fvSchemes.fvsc, else ``default``; the 3-D check is applied by the lookup."""
    d = dev.fvSchemes["fvsc"]
    return d[term_name] if term_name in d else d["default"]


def grad(dev, vf, out=None):
    return fvscStencil.lookupOrNew(fvscOpName(dev, f"grad({vf.name})"), dev).Grad(vf, out)


def div(dev, vf, out=None):
    return fvscStencil.lookupOrNew(fvscOpName(dev, f"div({vf.name})"), dev).Div(vf, out)


def _scheme_text(v):
    return " ".join(str(x) for x in v) if isinstance(v, (list, tuple)) else str(v)


def qgdInterpolate(dev, vf, out=None):
    """QGDInterpolate.H L38-67.  Without an entry (or with ``default none``) linearInterpolate; an ``interpolate(<name>)`` entry or
    any other default hands the field to fvc::interpolate -- with ``linear`` that is the same weights and the same numbers and
    takes the library path; any other scheme lives in OpenFOAM's scheme library and is refused."""
    schemes = dev.fvSchemes.get("interpolationSchemes", {})
    own = schemes.get(f"interpolate({vf.name})")
    word = _scheme_text(own) if own is not None else _scheme_text(schemes.get("default", "none"))
    if word not in ("linear",) + (() if own is not None else ("none",)):
        entry = f"interpolate({vf.name})" if own is not None else "default"
        raise L.QgdError(L.ERR_NOT_IMPLEMENTED, f"qgdInterpolate({vf.name}): interpolationSchemes{{{entry} {word};}} -- fvc::interpolate with a "
                                                f"scheme other than linear stays in OpenFOAM")
    m = dev.mesh
    nc = vf.ncomp
    if isinstance(vf, deviceVolField):
        L.check(L.lib.qgd_interpolate_dev(dev._h, nc, C.c_void_p(vf.internal), C.c_void_p(vf.boundary), C.c_void_p(out)), "qgd_interpolate_dev")
        return out
    out = np.zeros((m.nFaces, nc) if nc > 1 else (m.nFaces,))
    bnd = vf.boundary if vf.boundary.size else np.zeros(1)
    L.check(L.lib.qgd_interpolate(dev._h, nc, vf.internal.ctypes.data_as(L.c_double_p), bnd.ctypes.data_as(L.c_double_p),
                                  out.ctypes.data_as(L.c_double_p)), "qgd_interpolate")
    return out


def qgdFlux(dev, flux, psif, psi=None, flux_name=None):
    """QGDInterpolate.H L76-118: flux*psif unless ``divSchemes`` holds an entry for the flux's name (``div(<flux>,<psi>)``, L116); then
    fvc::flux(flux, psi, name): ``Gauss linear`` = the same numbers, ``Gauss upwind`` = flux times the upwind cell's psi
    (``qgd_flux_upwind``); limited schemes stay in OpenFOAM (refused).  ``default`` is not consulted (``found(fluxName)``, L86)."""
    flux = np.ascontiguousarray(flux, dtype=np.float64)
    psif = np.ascontiguousarray(psif, dtype=np.float64)
    nc = 1 if psif.ndim == 1 else psif.shape[1]
    out = np.zeros_like(psif)
    entry = dev.fvSchemes.get("divSchemes", {}).get(flux_name) if flux_name is not None else None
    word = _scheme_text(entry) if entry is not None else None
    if word == "Gauss upwind":
        if psi is None:
            raise ValueError("qgdFlux: the upwind branch needs the volField psi")
        bnd = psi.boundary if psi.boundary.size else np.zeros(1)
        L.check(L.lib.qgd_flux_upwind(dev._h, nc, flux.ctypes.data_as(L.c_double_p), psi.internal.ctypes.data_as(L.c_double_p),
                                      bnd.ctypes.data_as(L.c_double_p), out.ctypes.data_as(L.c_double_p)), "qgd_flux_upwind")
        return out
    if word not in (None, "Gauss linear"):
        raise L.QgdError(L.ERR_NOT_IMPLEMENTED, f"qgdFlux({flux_name}): divSchemes{{{flux_name} {word};}} -- fvc::flux with a scheme other "
                                                f"than Gauss linear / Gauss upwind stays in OpenFOAM")
    L.check(L.lib.qgd_flux(dev._h, nc, flux.ctypes.data_as(L.c_double_p), psif.ctypes.data_as(L.c_double_p),
                           out.ctypes.data_as(L.c_double_p)), "qgd_flux")
    return out


def device_field(dev, name):
    m = dev.mesh
    n = {"hQGD": m.nCells, "hQGDf": m.nFaces, "hQGD.boundary": m.nBoundaryFaces}[name]
    out = np.zeros(max(n, 1))
    L.check(L.lib.qgd_device_get(dev._h, name.encode(), out.ctypes.data_as(L.c_double_p), out.size), "qgd_device_get")
    return out[:n]
