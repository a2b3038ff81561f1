"""The OpenFOAM adapter (qgdsolver_amd/foam/hipStencil.{H,C}) cannot be built here -- there is no OpenFOAM in the image -- so it is
kept from rotting by a SYNTAX AND CONTROL-FLOW harness: tests/cpp/foam_stub/ holds the smallest stand-in declarations under which
the file parses (clearly not OpenFOAM, no numerics, not parity evidence), and tests/cpp/foam_adapter_harness.cpp drives the
adapter against a recording mock of the C-ABI.  Asserted: correctBoundaryConditions() only for the
GaussVolPoint word [GaussVolPointStencil_8C L73-121 vs reducedFaceNormalStencil_8C L69-108]; Pstream::parRun() and processor
patches with faces end in FatalError."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, "tests", "cpp", "foam_stub"), "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "qgdsolver_amd", "foam")]
SRC = os.path.join(ROOT, "qgdsolver_amd", "foam", "hipStencil.C")


def test_adapter_source_parses():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", *INC, SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapter_control_flow(tmp_path):
    exe = str(tmp_path / "foam_harness")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", *INC, SRC, os.path.join(ROOT, "tests", "cpp", "foam_adapter_harness.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "HARNESS OK" in r.stdout, r.stdout + r.stderr
    out = r.stdout
    # the three behaviours, by name, so a silently dropped check shows up
    assert "hipReduced: inputs' boundary conditions left as they are" in out
    assert "hipLeastSquares: inputs' boundary conditions left as they are" in out
    assert "hipGaussVolPoint: correctBoundaryConditions() before every operator" in out
    assert "ok   Pstream::parRun() -> FatalError" in out
    assert "ok   processor patch with faces -> FatalError" in out


def test_stub_is_labelled_as_a_harness():
    text = open(os.path.join(ROOT, "tests", "cpp", "foam_stub", "foamStub.H")).read()
    assert "NOT OpenFOAM" in text and "NOT parity evidence" in text
