"""A case directory written in OpenFOAM's ASCII formats runs through foamfile.load_case on the device and matches the
oracle set up by hand with the same numbers (the reader side of the adapter validation)."""
import os

import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import foamfile as ff
import cases
from oracle import OracleCase
from test_foamfile import write_step_case
from util import oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stencil", ["leastSquares", "GaussVolPoint"])
def test_case_directory_runs_like_the_hand_built_case(tmp_path, stencil):
    case_dir = str(tmp_path)
    mesh = write_step_case(case_dir, stencil)
    # non-uniform initial fields, written and read at full precision
    C = mesh.array("C").reshape(-1, 3)
    T0 = 1.0 + 0.05 * np.sin(2.0 * C[:, 0]) * np.cos(3.0 * C[:, 1])
    p0 = 1.0 + 0.05 * np.cos(1.5 * C[:, 0] + C[:, 1])
    _, _, _, bcs = ff.read_case_setup(case_dir)
    word = {"none": "empty"}
    for name, arr in (("T", T0), ("p", p0)):
        patches = {}
        for pn, bc in zip(mesh.patch_names, bcs):
            kind, val = bc[name]
            patches[pn] = (word.get(kind, kind), None if val is None else np.float64(val))
        ff.write_field(os.path.join(case_dir, "0", name), mesh, name, arr, patches)

    dev, gc = ff.load_case(case_dir)
    assert gc.options.deltaT == 5e-4 and gc.options.implicitDiffusion == 0

    oc = OracleCase(oracle_mesh_of(mesh), q.default_options(stencil=stencil, deltaT=5e-4, R=1 / 1.4, Cv=1 / 1.4 / 0.4, mu=0.0,
                                                            Pr=1.0, ScQGD=1.0, PrQGD=1.0, alphaQGD=0.5))
    cases.forward_step_bcs(oc)
    U0 = np.zeros((mesh.nCells, 3))
    U0[:, 0] = 3.0
    oc.set_fields(U0, T0, p0)
    gc.step(25)
    oc.step(25)
    for n in ("rho", "U", "p", "e"):
        assert rel_err(gc.field(n), oc.field(n)) <= 1e-10, (stencil, n)

    # runTime.write(): the time directory reads back to the same state bit for bit
    ff.write_time(gc, case_dir, "0.0125", bcs)
    m2 = gc.mesh
    for n in ("U", "T", "p", "rho"):
        vals, patches = ff.read_field(os.path.join(case_dir, "0.0125", n), m2)
        want = gc.field(n)
        assert np.array_equal(vals.reshape(want.shape), want), n
    assert patches["inlet"]["type"] == "calculated" or patches["inlet"]["type"] == "fixedValue"
    gc.close()
    dev.close()


def test_implicit_diffusion_default_runs(tmp_path):
    """an absent QGD.implicitDiffusion means true in the reference [QGDThermo.C L70-82]: the case runs the implicit branch"""
    case_dir = str(tmp_path)
    write_step_case(case_dir)
    tp = os.path.join(case_dir, "constant", "thermophysicalProperties")
    text = open(tp).read()
    open(tp, "w").write(text.replace("implicitDiffusion false;", ""))
    dev, gc = ff.load_case(case_dir)
    assert gc.options.implicitDiffusion == 1 and gc.thermo.implicitDiffusion() is True
    gc.step(3)
    assert gc.info()["minRho"] > 0
    gc.close(); dev.close()


def test_application_and_reference_run_comparison(tmp_path):
    """python -m qgdsolver_amd.QGDFoam -case ... writes time directories; scripts/compare_with_reference_run.py accepts a
    'reference run' (here: the oracle's fields written in OpenFOAM format) and rejects a perturbed one."""
    import subprocess
    import sys

    case_dir = str(tmp_path)
    mesh = write_step_case(case_dir, "GaussVolPoint")
    cd = os.path.join(case_dir, "system", "controlDict")
    open(cd, "a").write("writeControl timeStep;\nwriteInterval 10;\ntimePrecision 8;\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "qgdsolver_amd.QGDFoam", "-case", case_dir, "-nSteps", "20"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Time = 0.005" in r.stdout and "Time = 0.01" in r.stdout and r.stdout.rstrip().endswith("End")
    assert os.path.exists(os.path.join(case_dir, "0.005", "U")) and os.path.exists(os.path.join(case_dir, "0.01", "rho"))

    # the "reference run": the oracle advanced 20 steps, written as the time directory 0.01
    oc = OracleCase(oracle_mesh_of(mesh), q.default_options(stencil="GaussVolPoint", deltaT=5e-4, R=1 / 1.4, Cv=1 / 1.4 / 0.4, mu=0.0,
                                                            Pr=1.0))
    cases.forward_step_bcs(oc)
    U0 = np.zeros((mesh.nCells, 3))
    U0[:, 0] = 3.0
    oc.set_fields(U0, np.ones(mesh.nCells), np.ones(mesh.nCells))
    oc.step(20)
    ref_dir = os.path.join(case_dir, "0.01")
    calc = {pn: ("calculated", None) for pn in mesh.patch_names}
    for n in ("rho", "U", "p"):
        ff.write_field(os.path.join(ref_dir, n), mesh, n, oc.field(n), calc)
    ff.write_field(os.path.join(ref_dir, "T"), mesh, "T", oc.field("e") / (1 / 1.4 / 0.4), calc)
    script = os.path.join(root, "scripts", "compare_with_reference_run.py")
    r = subprocess.run([sys.executable, script, case_dir, "0", "0.01"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "PARITY OK" in r.stdout, r.stdout + r.stderr[-2000:]
    rho = oc.field("rho")
    rho[7] *= 1 + 1e-8
    ff.write_field(os.path.join(ref_dir, "rho"), mesh, "rho", rho, calc)
    r = subprocess.run([sys.executable, script, case_dir, "0", "0.01"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 1 and "PARITY FAILED" in r.stdout


@pytest.mark.parametrize("world,renumber,adjust,implicit", [(3, "morton", "no", False), (2, "rcm", "yes", False), (2, "morton", "no", True),
                                                            (3, "rcm", "yes", True)])
def test_application_on_several_ranks(tmp_path, world, renumber, adjust, implicit):
    """torch.distributed.run -m qgdsolver_amd.QGDFoam on W ranks (gloo-staged halo messages, all ranks on this one GPU),
    cells relabelled on the device side: the written time directory equals the single-rank run's to rounding."""
    import shutil
    import subprocess
    import sys

    one = str(tmp_path / "one")
    many = str(tmp_path / "many")
    os.makedirs(one)
    mesh = write_step_case(one, "GaussVolPoint")
    cd = os.path.join(one, "system", "controlDict")
    text = open(cd).read().replace("adjustTimeStep no;", f"adjustTimeStep {adjust};")
    open(cd, "w").write(text + "writeControl timeStep;\nwriteInterval 12;\ntimePrecision 10;\nmaxDeltaT 1;\n")
    if implicit:      # the reference's default branch (an absent QGD.implicitDiffusion means true): phases 20..35 over DistWorld
        tp = os.path.join(one, "constant", "thermophysicalProperties")
        tp_text = open(tp).read().replace("implicitDiffusion false;", "")
        open(tp, "w").write(tp_text)
    shutil.copytree(one, many)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "qgdsolver_amd.QGDFoam", "-case", one, "-nSteps", "12"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(29540 + world), "-m", "qgdsolver_amd.QGDFoam", "-case", many, "-nSteps", "12",
                        "-renumber", renumber, "-backend", "gloo"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert f"on {world} rank(s)" in r.stdout
    times = sorted(d for d in os.listdir(one) if d not in ("0", "constant", "system"))
    assert len(times) == 1 and os.path.isdir(os.path.join(many, times[0])), (times, os.listdir(many))
    for n in ("rho", "U", "p", "T"):
        a, _ = ff.read_field(os.path.join(one, times[0], n), mesh)
        b, _ = ff.read_field(os.path.join(many, times[0], n), mesh)
        assert np.abs(a - b).max() <= (1e-9 if implicit else 1e-12) * np.abs(a).max(), (n, np.abs(a - b).max())
