"""Host polyMesh handle (qgd_mesh_t): OpenFOAM-ordered flat arrays."""
import ctypes as C

import numpy as np

from . import _lib as L

_INT_ARRAYS = {"faceOffsets", "facePoints", "owner", "neighbour", "patchStart", "patchSize", "patchType",
               "haloGhost0", "haloGhost1", "haloSend0", "haloSend1"}


def _dp(a):
    return a.ctypes.data_as(L.c_double_p)


def _ip(a):
    return a.ctypes.data_as(L.c_int32_p)


class PolyMesh:
    def __init__(self, handle):
        self._h = handle
        s = (C.c_int64 * 7)()
        L.check(L.lib.qgd_mesh_sizes(self._h, s), "qgd_mesh_sizes")
        (self.nPoints, self.nFaces, self.nInternalFaces, self.nCells, self.nPatches, self.nFacePoints,
         self.nGeometricD) = [int(x) for x in s]
        self.nBoundaryFaces = self.nFaces - self.nInternalFaces

    # ---- constructors -------------------------------------------------------
    @classmethod
    def box(cls, nx, ny, nz, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0), patch_types=None, k_range=None):
        """blockMesh-numbered box; k_range=(kLo,kHi) builds that k-slab of the nz-tall box."""
        lo_a = np.asarray(lo, dtype=np.float64)
        hi_a = np.asarray(hi, dtype=np.float64)
        pt = np.asarray(patch_types if patch_types is not None else [L.PATCH_GENERIC] * 6, dtype=np.int32)
        k_lo, k_hi = k_range if k_range is not None else (0, nz)
        h = C.c_void_p()
        L.check(L.lib.qgd_mesh_box(nx, ny, nz, k_lo, k_hi, _dp(lo_a), _dp(hi_a), _ip(pt), C.byref(h)), "qgd_mesh_box")
        return cls(h)

    @classmethod
    def forward_step(cls, nx, ny, ix_step, iy_step, lx=3.0, ly=1.0, lz=0.1):
        h = C.c_void_p()
        L.check(L.lib.qgd_mesh_forward_step(nx, ny, ix_step, iy_step, lx, ly, lz, C.byref(h)), "qgd_mesh_forward_step")
        return cls(h)

    @classmethod
    def from_arrays(cls, points, face_offsets, face_points, owner, neighbour, n_cells, patch_start, patch_size, patch_type):
        points = np.ascontiguousarray(points, dtype=np.float64)
        fo = np.ascontiguousarray(face_offsets, dtype=np.int32)
        fp = np.ascontiguousarray(face_points, dtype=np.int32)
        ow = np.ascontiguousarray(owner, dtype=np.int32)
        ne = np.ascontiguousarray(neighbour, dtype=np.int32)
        ps = np.ascontiguousarray(patch_start, dtype=np.int32)
        pz = np.ascontiguousarray(patch_size, dtype=np.int32)
        pt = np.ascontiguousarray(patch_type, dtype=np.int32)
        h = C.c_void_p()
        L.check(L.lib.qgd_mesh_create(points.size // 3, _dp(points), ow.size, _ip(fo), _ip(fp), ne.size, _ip(ow), _ip(ne),
                                      int(n_cells), ps.size, _ip(ps), _ip(pz), _ip(pt), C.byref(h)), "qgd_mesh_create")
        return cls(h)

    # ---- access ---------------------------------------------------------------
    def array(self, name):
        n = {
            "points": 3 * self.nPoints, "faceOffsets": self.nFaces + 1, "facePoints": self.nFacePoints,
            "owner": self.nFaces, "neighbour": self.nInternalFaces, "patchStart": self.nPatches,
            "patchSize": self.nPatches, "patchType": self.nPatches, "Sf": 3 * self.nFaces, "magSf": self.nFaces,
            "Cf": 3 * self.nFaces, "C": 3 * self.nCells, "V": self.nCells, "weights": self.nFaces,
            "deltaCoeffs": self.nFaces, "nonOrthDeltaCoeffs": self.nFaces,
            "haloGhost0": self.nCells, "haloGhost1": self.nCells, "haloSend0": self.nCells, "haloSend1": self.nCells,
        }[name]
        dt = np.int32 if name in _INT_ARRAYS else np.float64
        out = np.full(max(n, 1), -1 if dt == np.int32 else 0, dtype=dt)
        L.check(L.lib.qgd_mesh_get(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes), f"qgd_mesh_get({name})")
        if name.startswith("halo"):
            return out[out >= 0]
        return out[:n]

    def primitives(self):
        """Arrays in the order qgd_mesh_create / orc_mesh_create take them."""
        return dict(points=self.array("points"), faceOffsets=self.array("faceOffsets"), facePoints=self.array("facePoints"),
                    owner=self.array("owner"), neighbour=self.array("neighbour"), nCells=self.nCells,
                    patchStart=self.array("patchStart"), patchSize=self.array("patchSize"), patchType=self.array("patchType"))

    def jitter(self, amplitude, seed=2024):
        L.check(L.lib.qgd_mesh_jitter(self._h, float(amplitude), int(seed)), "qgd_mesh_jitter")
        return self

    def split_quads(self, stride):
        L.check(L.lib.qgd_mesh_split_quads(self._h, int(stride)), "qgd_mesh_split_quads")
        self.__init__(self._h)
        return self

    def split_edges(self, stride):
        L.check(L.lib.qgd_mesh_split_edges(self._h, int(stride)), "qgd_mesh_split_edges")
        self.__init__(self._h)
        return self

    def close(self):
        if self._h:
            L.lib.qgd_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
