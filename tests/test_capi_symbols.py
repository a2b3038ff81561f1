"""CPU: the C-ABI library loads and exports every symbol include/qgd_amd.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qgd_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qgd_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L.lib, n), f"libqgd_amd.so does not export {n}"
    # and the ctypes table binds exactly the declared set
    assert sorted(L.SIGNATURES) == names


def test_version_and_error_text():
    assert b"gfx950" in L.lib.qgd_version()
    assert isinstance(L.lib.qgd_last_error(), bytes)


def test_no_oracle_linked_into_product():
    """The product library must not link or reference anything under oracle/."""
    data = open(L.LIB_PATH, "rb").read()
    assert b"orc_case" not in data and b"qgd_oracle" not in data
    for src in os.listdir(os.path.join(ROOT, "qgdsolver_amd", "csrc")):
        if src.endswith((".cpp", ".hpp", ".hip")):
            assert "oracle" not in open(os.path.join(ROOT, "qgdsolver_amd", "csrc", src)).read().lower(), src
    for src in os.listdir(os.path.join(ROOT, "qgdsolver_amd")):
        if src.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "qgdsolver_amd", src)).read().lower(), src


def test_device_entries_fail_loudly_without_gpu():
    if q.device_count() > 0:
        pytest.skip("a GPU is visible here")
    mesh = q.PolyMesh.box(3, 3, 3)
    with pytest.raises(q.QgdError) as ei:
        q.Device(mesh)
    assert ei.value.code == L.ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_bad_arguments_return_status():
    h = C.c_void_p()
    rc = L.lib.qgd_mesh_box(0, 3, 3, 0, 3, None, None, None, C.byref(h))
    assert rc == L.ERR_INVALID
    import numpy as np
    lo = np.zeros(3); hi = np.ones(3)
    rc = L.lib.qgd_mesh_box(0, 3, 3, 0, 3, lo.ctypes.data_as(L.c_double_p), hi.ctypes.data_as(L.c_double_p), None, C.byref(h))
    assert rc == L.ERR_INVALID and b"makeBox" in L.lib.qgd_last_error()
