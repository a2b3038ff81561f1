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


def test_header_is_plain_c_and_links(tmp_path):
    """include/qgd_amd.h is a C header (C99, -pedantic -Werror) and a C program links against the library: the boundary a
    non-C++ host (cgo, JNI, ctypes, Fortran bind(C)) would use"""
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include "qgd_amd.h"\n'
                   'int main(void) {\n'
                   '    qgd_case_options o; qgd_poisson_control c; qgd_mesh_t m = 0; int64_t s[7];\n'
                   '    double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1}; int32_t pt[6] = {0, 0, 0, 0, 0, 0};\n'
                   '    if (qgd_case_options_default(&o) || qgd_poisson_control_default(&c)) return 1;\n'
                   '    if (qgd_mesh_box(3, 2, 2, 0, 2, lo, hi, pt, &m) || qgd_mesh_sizes(m, s)) return 2;\n'
                   '    printf("%s %lld %d\\n", qgd_version(), (long long)s[3], o.stencil);\n'
                   '    return qgd_mesh_free(m);\n}\n')
    exe = tmp_path / "t"
    libdir = os.path.join(ROOT, "qgdsolver_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                        "-o", str(exe), "-L", libdir, "-lqgd_amd", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[-2:] == ["12", "2"], (r.returncode, r.stdout, r.stderr)


def test_native_handle_is_freed_once_by_whoever_comes_first():
    """fvsc.Device holds the handles of the cases created on it and frees those still open before itself (a case freed after its
    device would read freed memory); the case's own close() then finds nothing left to do."""
    from qgdsolver_amd._lib import NativeHandle
    freed = []
    h = NativeHandle(1234, freed.append)
    assert h.value == 1234
    h.free(); h.free()
    assert freed == [1234] and h.value is None
