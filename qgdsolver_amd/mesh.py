"""Host polyMesh handle (qgd_mesh_t): OpenFOAM-ordered flat arrays."""
import ctypes as C

import numpy as np

from . import _lib as L

_INT_ARRAYS = {"degenerateFaces", "faceOffsets", "facePoints", "owner", "neighbour", "patchStart", "patchSize", "patchType",
               "haloPeer", "haloSelf", "cellGlobal", "faceGlobal", "pointGlobal"}


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
        """named array of qgd_mesh_get (sizes are asked from the library)"""
        nbytes = C.c_int64()
        L.check(L.lib.qgd_mesh_get(self._h, name.encode(), C.byref(nbytes), -1), f"qgd_mesh_get({name})")
        integer = name in _INT_ARRAYS or name.startswith("haloGhost") or name.startswith("haloSend")
        dt = np.int32 if integer else np.float64
        out = np.zeros(nbytes.value // np.dtype(dt).itemsize, dtype=dt)
        if out.size:
            L.check(L.lib.qgd_mesh_get(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes), f"qgd_mesh_get({name})")
        return out

    @property
    def halo_slots(self):
        n = C.c_int32()
        L.check(L.lib.qgd_mesh_halo_slots(self._h, C.byref(n)), "qgd_mesh_halo_slots")
        return n.value

    def primitives(self):
        """Arrays in the order qgd_mesh_create / orc_mesh_create take them."""
        return dict(points=self.array("points"), faceOffsets=self.array("faceOffsets"), facePoints=self.array("facePoints"),
                    owner=self.array("owner"), neighbour=self.array("neighbour"), nCells=self.nCells,
                    patchStart=self.array("patchStart"), patchSize=self.array("patchSize"), patchType=self.array("patchType"))

    def set_geometry(self, Sf, Cf, C_, V):
        """hand over the caller's own face area vectors / centres, cell centres / volumes (qgd_mesh_set_geometry: what the
        OpenFOAM adapter does with mesh.Sf(), Cf(), C(), V()); derived coefficients are recomputed from them"""
        a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (Sf, Cf, C_, V)]
        assert a[0].size == 3 * self.nFaces and a[1].size == 3 * self.nFaces and a[2].size == 3 * self.nCells and a[3].size == self.nCells
        L.check(L.lib.qgd_mesh_set_geometry(self._h, *[_dp(x) for x in a]), "qgd_mesh_set_geometry")
        self.__init__(self._h)
        return self

    def set_degenerate_faces(self, faces):
        """the faceSet degenerateStencilFaces of the leastSquares stencil (leastSquaresStencil.C L58-133): internal faces whose
        gradient falls back to nf * snGrad; before the Device is created"""
        f = np.ascontiguousarray(faces, dtype=np.int32)
        L.check(L.lib.qgd_mesh_set_degenerate_faces(self._h, int(f.size), f.ctypes.data_as(L.c_int32_p) if f.size else None),
                "qgd_mesh_set_degenerate_faces")
        return self

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

    # ---- renumbering / cell-range sharding ---------------------------------------
    def rcm_order(self):
        """newOfOld of a reverse Cuthill-McKee ordering (bandwidth reduction before cutting cell ranges)"""
        out = np.zeros(self.nCells, dtype=np.int32)
        L.check(L.lib.qgd_mesh_rcm_order(self._h, _ip(out)), "qgd_mesh_rcm_order")
        return out

    def morton_order(self):
        """newOfOld of the Morton (Z-curve) order of the cell centres"""
        out = np.zeros(self.nCells, dtype=np.int32)
        L.check(L.lib.qgd_mesh_morton_order(self._h, _ip(out)), "qgd_mesh_morton_order")
        return out

    def renumber(self, new_of_old):
        """relabel the cells in place; returns faceNewOfOld (new face label, or -1-label where the face was reversed)"""
        perm = np.ascontiguousarray(new_of_old, dtype=np.int32)
        assert perm.size == self.nCells
        face_map = np.zeros(self.nFaces, dtype=np.int32)
        L.check(L.lib.qgd_mesh_renumber(self._h, _ip(perm), _ip(face_map)), "qgd_mesh_renumber")
        self.__init__(self._h)
        return face_map

    def shard(self, n_ranks, rank, cell_start=None):
        """the shard of ``rank``: owned cell range + one vertex-connected ghost layer + halo slots per neighbour"""
        if cell_start is None:
            cell_start = [(self.nCells * r) // n_ranks for r in range(n_ranks + 1)]
        cs = np.ascontiguousarray(cell_start, dtype=np.int32)
        assert cs.size == n_ranks + 1
        h = C.c_void_p()
        L.check(L.lib.qgd_mesh_shard(self._h, int(n_ranks), _ip(cs), int(rank), C.byref(h)), "qgd_mesh_shard")
        out = PolyMesh(h)
        names = getattr(self, "patch_names", None)
        if names:
            out.patch_names = list(names) + (["halo"] if n_ranks > 1 else [])
        return out

    def unroll_cyclic(self, pairs=None):
        """translational cyclic patch pairs served by ghost cells (qgd_mesh_unroll_cyclic): this mesh + one layer of translated copies of its
        own cells behind every half, the pairs' faces glued into internal faces; pairs = [(patchA, patchB), ...], None pairs consecutive
        cyclic patches.  A QGDFoamCase on the result steps with plain step(): the library refreshes the copies itself."""
        flat = np.ascontiguousarray([x for pr in (pairs or []) for x in pr], dtype=np.int32)
        h = C.c_void_p()
        L.check(L.lib.qgd_mesh_unroll_cyclic(self._h, flat.size // 2, _ip(flat) if flat.size else None, C.byref(h)), "qgd_mesh_unroll_cyclic")
        out = PolyMesh(h)
        names = getattr(self, "patch_names", None)
        if names:
            out.patch_names = list(names) + ["halo"]
        return out

    def close(self):
        if self._h:
            L.lib.qgd_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
