"""ctypes binding of libqgd_amd.so (the C-ABI declared in include/qgd_amd.h).

The shared library is built in-tree by ``qgdsolver_amd/csrc/Makefile`` (or
``__graft_entry__.build()``).  There is no Python or CPU fallback: if the
library is missing, importing this module raises.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# QGD_AMD_LIB: another build of the same library (A/B timing of compile-time variants, scripts/ab_variants.sh)
LIB_PATH = os.environ.get("QGD_AMD_LIB") or os.path.join(_HERE, "libqgd_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C qgdsolver_amd/csrc). "
        "qgdsolver_amd has no CPU fallback."
    )

# torch ships its own copy of the HIP runtime; when both live in one process torch's has to be loaded first (otherwise
# the second runtime sees no devices).  Multi-rank runs always use torch.distributed, so load it before the library.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "torch" not in sys.modules:
    try:
        import torch  # noqa: F401
    except ImportError:
        pass

lib = C.CDLL(LIB_PATH)

c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_double_p = C.POINTER(C.c_double)
handle = C.c_void_p
handle_p = C.POINTER(C.c_void_p)


class CaseOptions(C.Structure):
    """qgd_case_options of include/qgd_amd.h."""

    _fields_ = [
        ("stencil", C.c_int32), ("implicitDiffusion", C.c_int32), ("adjustTimeStep", C.c_int32), ("consistentEnergy", C.c_int32),
        ("R", C.c_double), ("Cv", C.c_double), ("mu", C.c_double), ("Pr", C.c_double), ("ScQGD", C.c_double),
        ("PrQGD", C.c_double), ("alphaQGD", C.c_double), ("deltaT", C.c_double), ("maxCo", C.c_double),
        ("maxDeltaT", C.c_double), ("cTau", C.c_double), ("implicitTol", C.c_double), ("implicitMaxIter", C.c_int32),
        ("fluxSchemeU", C.c_int32), ("fluxSchemeH", C.c_int32), ("pad_", C.c_int32), ("termStencil", C.c_int32 * 4),
    ]


class QhdOptions(C.Structure):
    """qgd_qhd_options"""
    _fields_ = [("stencil", C.c_int32), ("implicitDiffusion", C.c_int32), ("tauModel", C.c_int32), ("pRefCell", C.c_int32),
                ("pMaxIter", C.c_int32), ("precond", C.c_int32),
                ("rho0", C.c_double), ("mu", C.c_double), ("Pr", C.c_double), ("beta", C.c_double), ("g", C.c_double * 3),
                ("deltaT", C.c_double), ("Tau", C.c_double), ("aQGD", C.c_double), ("UQHD", C.c_double), ("T0", C.c_double),
                ("Gr", C.c_double), ("pTol", C.c_double), ("pRelTol", C.c_double), ("pRefValue", C.c_double),
                ("implicitTol", C.c_double), ("implicitMaxIter", C.c_int32),
                ("fluxSchemeU", C.c_int32), ("fluxSchemeT", C.c_int32), ("pad_", C.c_int32)]


# every symbol include/qgd_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "qgd_version": (C.c_char_p, []),
    "qgd_last_error": (C.c_char_p, []),
    "qgd_device_count": (C.c_int, []),
    "qgd_mesh_create": (C.c_int, [C.c_int32, c_double_p, C.c_int32, c_int32_p, c_int32_p, C.c_int32, c_int32_p, c_int32_p,
                                  C.c_int32, C.c_int32, c_int32_p, c_int32_p, c_int32_p, handle_p]),
    "qgd_mesh_box": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, c_int32_p, handle_p]),
    "qgd_mesh_forward_step": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, handle_p]),
    "qgd_mesh_jitter": (C.c_int, [handle, C.c_double, C.c_uint64]),
    "qgd_mesh_split_quads": (C.c_int, [handle, C.c_int32]),
    "qgd_mesh_split_edges": (C.c_int, [handle, C.c_int32]),
    "qgd_mesh_renumber": (C.c_int, [handle, c_int32_p, c_int32_p]),
    "qgd_mesh_rcm_order": (C.c_int, [handle, c_int32_p]),
    "qgd_mesh_morton_order": (C.c_int, [handle, c_int32_p]),
    "qgd_mesh_shard": (C.c_int, [handle, C.c_int32, c_int32_p, C.c_int32, handle_p]),
    "qgd_mesh_unroll_cyclic": (C.c_int, [handle, C.c_int32, c_int32_p, handle_p]),
    "qgd_mesh_halo_slots": (C.c_int, [handle, c_int32_p]),
    "qgd_mesh_set_geometry": (C.c_int, [handle, c_double_p, c_double_p, c_double_p, c_double_p]),
    "qgd_mesh_set_degenerate_faces": (C.c_int, [handle, C.c_int32, c_int32_p]),
    "qgd_mesh_free": (C.c_int, [handle]),
    "qgd_mesh_sizes": (C.c_int, [handle, c_int64_p]),
    "qgd_mesh_get": (C.c_int, [handle, C.c_char_p, C.c_void_p, C.c_int64]),
    "qgd_device_create": (C.c_int, [handle, C.c_int, handle_p]),
    "qgd_device_create_with": (C.c_int, [handle, C.c_int, C.c_int32, handle_p]),
    "qgd_device_fused_blocks": (C.c_int, [handle, C.POINTER(C.c_int64)]),
    "qgd_device_free": (C.c_int, [handle]),
    "qgd_stencil_lookup": (C.c_int, [handle, C.c_char_p, C.POINTER(C.c_int)]),
    "qgd_fvsc_grad_s": (C.c_int, [handle, C.c_int, c_double_p, c_double_p, c_double_p]),
    "qgd_fvsc_grad_v": (C.c_int, [handle, C.c_int, c_double_p, c_double_p, c_double_p]),
    "qgd_fvsc_div_v": (C.c_int, [handle, C.c_int, c_double_p, c_double_p, c_double_p]),
    "qgd_fvsc_div_t": (C.c_int, [handle, C.c_int, c_double_p, c_double_p, c_double_p]),
    "qgd_fvsc_grad_s_dev": (C.c_int, [handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qgd_fvsc_grad_v_dev": (C.c_int, [handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qgd_fvsc_div_v_dev": (C.c_int, [handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qgd_fvsc_div_t_dev": (C.c_int, [handle, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qgd_interpolate_dev": (C.c_int, [handle, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qgd_device_sync": (C.c_int, [handle]),
    "qgd_device_copy": (C.c_int, [handle, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "qgd_qhd_fluxes_dev": (C.c_int, [handle, C.c_int, C.c_void_p, C.c_void_p]),
    "qgd_species_flux_dev": (C.c_int, [handle, C.c_int] + [C.c_void_p] * 10),
    "qgd_device_op_times": (C.c_int, [handle, c_double_p]),
    "qgd_device_face_tiles": (C.c_int, [handle, C.POINTER(C.c_int64)]),
    "qgd_case_fused_info": (C.c_int, [handle, C.POINTER(C.c_int64)]),
    "qgd_interpolate": (C.c_int, [handle, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "qgd_flux": (C.c_int, [handle, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "qgd_flux_upwind": (C.c_int, [handle, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p]),
    "qgd_device_get": (C.c_int, [handle, C.c_char_p, c_double_p, C.c_int64]),
    "qgd_qhd_fluxes": (C.c_int, [handle, C.c_int, C.c_void_p, C.c_void_p]),
    "qgd_qhd_options_default": (C.c_int, [C.POINTER(QhdOptions)]),
    "qgd_qhd_case_create": (C.c_int, [handle, C.POINTER(QhdOptions), handle_p]),
    "qgd_qhd_case_free": (C.c_int, [handle]),
    "qgd_qhd_case_set_bc": (C.c_int, [handle, C.c_int32, C.c_int32, c_double_p, C.c_int32, C.c_double, C.c_int32, C.c_double]),
    "qgd_qhd_case_set_fields": (C.c_int, [handle, c_double_p, c_double_p, c_double_p]),
    "qgd_qhd_case_step": (C.c_int, [handle, C.c_int32]),
    "qgd_qhd_case_get_field": (C.c_int, [handle, C.c_char_p, c_double_p, C.c_int64]),
    "qgd_qhd_case_info": (C.c_int, [handle, c_double_p]),
    "qgd_qhd_case_fused_info": (C.c_int, [handle, C.POINTER(C.c_int64)]),
    "qgd_qhd_case_implicit_info": (C.c_int, [handle, c_double_p]),
    "qgd_qhd_case_implicit_control": (C.c_int, [handle, c_double_p, C.c_int]),
    "qgd_qhd_case_implicit_control_ptr": (C.c_int, [handle, handle_p]),
    "qgd_qhd_case_implicit_solve_status": (C.c_int, [handle, c_double_p]),
    "qgd_qhd_case_step_phase": (C.c_int, [handle, C.c_int]),
    "qgd_species_step": (C.c_int, [handle] + [c_double_p] * 6 + [C.c_double, C.c_double, c_double_p, c_double_p, c_double_p]),
    "qgd_species_step_dev": (C.c_int, [handle] + [C.c_void_p] * 6 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qgd_species_step_implicit": (C.c_int, [handle, c_double_p, c_double_p, C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_double,
                                            C.c_double, c_double_p, C.c_double, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "qgd_species_step_implicit_dev": (C.c_int, [handle, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                C.c_double, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, c_double_p]),
    "qgd_device_lsq_stencil": (C.c_int, [handle, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    "qgd_qhd_case_pending": (C.c_int, [handle, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), c_int64_p]),
    "qgd_qhd_case_control_ptr": (C.c_int, [handle, C.POINTER(C.c_void_p)]),
    "qgd_qhd_case_solve_status": (C.c_int, [handle, c_double_p]),
    "qgd_qhd_case_control": (C.c_int, [handle, c_double_p, C.c_int]),
    "qgd_qhd_case_sync": (C.c_int, [handle]),
    "qgd_qhd_case_sweep_time": (C.c_int, [handle, C.c_int, c_double_p]),
    "qgd_case_implicit_apply_time": (C.c_int, [handle, C.c_int, c_double_p]),
    "qgd_qhd_case_halo_count": (C.c_int, [handle, C.c_int, C.c_int, c_int64_p, c_int64_p]),
    "qgd_qhd_case_halo_pack": (C.c_int, [handle, C.c_int, C.c_int, C.c_void_p]),
    "qgd_qhd_case_halo_unpack": (C.c_int, [handle, C.c_int, C.c_int, C.c_void_p]),
    "qgd_qhd_case_halo_exchange": (C.c_int, [handle, handle, c_int32_p, C.c_int, C.c_int]),
    "qgd_qhd_case_step_sharded": (C.c_int, [handle, handle, c_int32_p, C.c_int, C.c_int32]),
    "qgd_case_options_default": (C.c_int, [C.POINTER(CaseOptions)]),
    "qgd_case_create": (C.c_int, [handle, C.POINTER(CaseOptions), handle_p]),
    "qgd_case_free": (C.c_int, [handle]),
    "qgd_case_set_bc": (C.c_int, [handle, C.c_int32, C.c_int32, c_double_p, C.c_int32, C.c_double, C.c_int32, C.c_double]),
    "qgd_case_set_fields": (C.c_int, [handle, c_double_p, c_double_p, c_double_p]),
    "qgd_case_set_qgd_coeffs": (C.c_int, [handle, c_double_p, c_double_p, c_double_p, c_double_p]),
    "qgd_case_update_fluxes": (C.c_int, [handle]),
    "qgd_case_step": (C.c_int, [handle, C.c_int32]),
    "qgd_case_get_field": (C.c_int, [handle, C.c_char_p, c_double_p, C.c_int64]),
    "qgd_case_info": (C.c_int, [handle, c_double_p]),
    "qgd_case_implicit_info": (C.c_int, [handle, c_double_p]),
    "qgd_struct_sizes": (C.c_int, [c_int64_p]),
    "qgd_case_implicit_halo_count": (C.c_int, [handle, C.c_int, C.c_int, c_int64_p, c_int64_p]),
    "qgd_case_implicit_halo_pack": (C.c_int, [handle, C.c_int, C.c_int, C.c_void_p]),
    "qgd_case_implicit_halo_unpack": (C.c_int, [handle, C.c_int, C.c_int, C.c_void_p]),
    "qgd_case_implicit_control": (C.c_int, [handle, c_double_p, C.c_int]),
    "qgd_case_implicit_control_ptr": (C.c_int, [handle, C.POINTER(C.c_void_p)]),
    "qgd_case_implicit_solve_status": (C.c_int, [handle, c_double_p]),
    "qgd_device_alloc": (C.c_int, [handle, C.c_int64, C.POINTER(C.c_void_p)]),
    "qgd_device_release": (C.c_int, [handle, C.c_void_p]),
    "qgd_species_flux": (C.c_int, [handle, C.c_int] + [c_double_p] * 10),
    "qgd_poisson_control_default": (C.c_int, [C.c_void_p]),
    "qgd_qhd_pressure": (C.c_int, [handle, c_double_p, c_double_p, c_double_p, c_int32_p, c_double_p, c_double_p, C.c_void_p,
                                   c_double_p, c_double_p, c_double_p]),
    "qgd_case_halo_count": (C.c_int, [handle, C.c_int, c_int64_p]),
    "qgd_case_halo_recv_count": (C.c_int, [handle, C.c_int, c_int64_p]),
    "qgd_case_mid_exchange_needed": (C.c_int, [handle, C.POINTER(C.c_int32)]),
    "qgd_case_mid_halo_count": (C.c_int, [handle, C.c_int, c_int64_p, c_int64_p]),
    "qgd_case_mid_halo_pack": (C.c_int, [handle, C.c_int, C.c_void_p]),
    "qgd_case_mid_halo_unpack": (C.c_int, [handle, C.c_int, C.c_void_p]),
    "qgd_case_halo_pack": (C.c_int, [handle, C.c_int, C.c_void_p]),
    "qgd_case_halo_unpack": (C.c_int, [handle, C.c_int, C.c_void_p]),
    "qgd_case_stream_sync": (C.c_int, [handle]),
    "qgd_case_set_stream": (C.c_int, [handle, C.c_void_p]),
    "qgd_case_step_phase": (C.c_int, [handle, C.c_int]),
    "qgd_case_set_halo_stream": (C.c_int, [handle, C.c_void_p]),
    "qgd_case_reduction_ptr": (C.c_int, [handle, C.POINTER(C.c_void_p)]),
    "qgd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "qgd_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, handle_p]),
    "qgd_comm_free": (C.c_int, [handle]),
    "qgd_comm_info": (C.c_int, [handle, c_int32_p]),
    "qgd_case_halo_exchange": (C.c_int, [handle, handle, c_int32_p, C.c_int]),
    "qgd_case_allreduce_max": (C.c_int, [handle, handle]),
    "qgd_case_step_sharded": (C.c_int, [handle, handle, c_int32_p, C.c_int, C.c_int]),
    "qgd_case_timing": (C.c_int, [handle, C.c_int]),
    "qgd_case_kernel_time": (C.c_int, [handle, C.c_int, c_double_p, c_int64_p]),
    "qgd_case_timing_reset": (C.c_int, [handle]),
    "qgd_case_device_bytes": (C.c_int, [handle, c_int64_p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

# the structs above must be the library's own (they carry no size member)
_sizes = (C.c_int64 * 4)()
lib.qgd_struct_sizes(_sizes)
if (_sizes[0], _sizes[1]) != (C.sizeof(CaseOptions), C.sizeof(QhdOptions)):
    raise ImportError(f"{LIB_PATH}: options structs of the library ({_sizes[0]}, {_sizes[1]} bytes) differ from this binding "
                      f"({C.sizeof(CaseOptions)}, {C.sizeof(QhdOptions)}): rebuild qgdsolver_amd/csrc")
ABI_VERSION = int(_sizes[3])

# enums of include/qgd_amd.h
QGD_OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_SCHEME, ERR_UNKNOWN_NAME, ERR_NOT_IMPLEMENTED = -1, -2, -3, -4, -5, -6
PATCH_GENERIC, PATCH_EMPTY, PATCH_SYMMETRYPLANE, PATCH_SYMMETRY, PATCH_WEDGE, PATCH_CYCLIC, PATCH_HALO = range(7)
DEVICE_NO_FUSED_TABLES, DEVICE_FUSED_ANY_BLOCKS = 1, 2   # flags of qgd_device_create_with
BC_ZEROGRADIENT, BC_FIXEDVALUE, BC_SLIP, BC_QGDFLUX, BC_NONE, BC_QHDFLUX = range(6)
FVSC_REDUCED, FVSC_LEASTSQUARES, FVSC_GAUSSVOLPOINT = range(3)
FLUX_LINEAR, FLUX_UPWIND = range(2)
K_POINT, K_FACE, K_BFACE, K_CELL, K_BC, K_BPOINT = range(6)


class QgdError(RuntimeError):
    """A non-zero status from the C-ABI (the adapter turns these into FatalError)."""

    def __init__(self, code, where):
        self.code = code
        msg = lib.qgd_last_error().decode()
        super().__init__(f"{where}: status {code}: {msg}")


def check(code, where):
    if code != QGD_OK:
        raise QgdError(code, where)


class NativeHandle:
    """an opaque library handle and the entry that frees it; free() is idempotent, so a case and its device may both call it"""

    def __init__(self, value, free_fn):
        self.value, self._free = value, free_fn

    def free(self):
        if self.value:
            self._free(self.value)
            self.value = None
