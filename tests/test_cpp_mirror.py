"""The C++ host-side mirror (include/qgd_amd_fvsc.hpp) compiles with plain g++ against the C-ABI (CPU) and runs
against the GPU library (gpu)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "fvsc_mirror_test.cpp")
LIBDIR = os.path.join(ROOT, "qgdsolver_amd")


def build(out):
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", out, "-L", LIBDIR, "-lqgd_amd",
           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_cpp_mirror_compiles_and_links(tmp_path):
    exe = str(tmp_path / "fvsc_mirror_test")
    build(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode in (0, 77), r.stdout + r.stderr  # 77 = no device here


@pytest.mark.gpu
def test_cpp_mirror_runs_on_gpu(tmp_path):
    exe = str(tmp_path / "fvsc_mirror_test")
    build(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
