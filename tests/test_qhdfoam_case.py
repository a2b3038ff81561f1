"""A QHDFoam case directory (buoyant cavity) written the way a user of the reference would write it: read by
foamfile.read_qhd_case_setup [QHDFoam/createFields.H L32-167], run by ``python -m qgdsolver_amd.QHDFoam -case <dir>``
[QHDFoam.C L63-139] with the reference's default implicitDiffusion (absent from the dictionary = true, QGDThermo.C L70-82), and
equal to the oracle set up by hand with the same numbers (VERDICT r03 "missing" #3)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L, foamfile as ff

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = "FoamFile {{ version 2.0; format ascii; class {cls}; object {obj}; }}\n"


def write_cavity_case(case_dir, stencil="GaussVolPoint", n=(8, 7, 6), implicit_line="", closure="HbyUQHD", jitter=0.1):
    """buoyant cavity: hot wall xMin, cold wall xMax (fixedValue T), adiabatic no-slip walls elsewhere, p qhdFlux on the walls"""
    mesh = q.PolyMesh.box(*n)
    if jitter:
        mesh.jitter(jitter, seed=7)
    mesh.patch_names = ["hot", "cold", "floor", "ceiling", "front", "back"]
    ff.write_polymesh(mesh, os.path.join(case_dir, "constant", "polyMesh"))

    def bf(entries):
        return "boundaryField\n{\n" + "\n".join(f"    {n} {{ {entries.get(n, entries['default'])} }}" for n in mesh.patch_names) + "\n}\n"

    os.makedirs(os.path.join(case_dir, "0"))
    os.makedirs(os.path.join(case_dir, "system"))
    with open(os.path.join(case_dir, "0", "U"), "w") as f:
        f.write(HDR.format(cls="volVectorField", obj="U") + "dimensions [0 1 -1 0 0 0 0];\ninternalField uniform (0 0 0);\n" +
                bf({"default": "type fixedValue; value uniform (0 0 0);"}))
    with open(os.path.join(case_dir, "0", "T"), "w") as f:
        f.write(HDR.format(cls="volScalarField", obj="T") + "dimensions [0 0 0 1 0 0 0];\ninternalField uniform 300;\n" +
                bf({"hot": "type fixedValue; value uniform 310;", "cold": "type fixedValue; value uniform 290;", "default": "type zeroGradient;"}))
    with open(os.path.join(case_dir, "0", "p"), "w") as f:
        f.write(HDR.format(cls="volScalarField", obj="p") + "dimensions [0 2 -2 0 0 0 0];\ninternalField uniform 0;\n" +
                bf({"ceiling": "type fixedGradient; gradient uniform 0.02;", "default": "type qhdFlux;"}))
    dicts = {"constTau": "constTauDict { Tau 2e-3; }", "HbyUQHD": "HbyUQHDDict { UQHD 0.8; }", "T0byGr": "T0byGrDict { T0 1.5; Gr 400; }",
             "H2bynuQHD": ""}
    with open(os.path.join(case_dir, "constant", "thermophysicalProperties"), "w") as f:
        f.write(HDR.format(cls="dictionary", obj="thermophysicalProperties") + textwrap.dedent(f'''
            thermoType {{ type heRhoQGDThermo; mixture pureMixture; transport const; thermo hConst;
                         equationOfState rhoConst; specie specie; energy sensibleInternalEnergy; }}
            mixture
            {{
                specie {{ molWeight 28.9; }}
                equationOfState {{ rho 1.2; }}
                thermodynamics {{ Cp 1005; Hf 0; }}
                transport {{ mu 1.8e-2; Pr 0.71; beta 3.4e-3; }}
            }}
            QGD
            {{
                {implicit_line}
                QGDCoeffs {closure};
                {dicts[closure]}
                pRefCell 17;
                pRefValue 0.25;
            }}
            '''))
    with open(os.path.join(case_dir, "constant", "gravitationalProperties"), "w") as f:
        f.write(HDR.format(cls="uniformDimensionedVectorField", obj="gravitationalProperties") + "g g [0 1 -2 0 0 0 0] (0 -9.81 0);\n")
    with open(os.path.join(case_dir, "system", "fvSchemes"), "w") as f:
        f.write(HDR.format(cls="dictionary", obj="fvSchemes") + f"ddtSchemes {{ default Euler; }}\ngradSchemes {{ default Gauss linear; }}\n"
                f"divSchemes {{ default none; }}\nlaplacianSchemes {{ default Gauss linear uncorrected; }}\n"
                f"interpolationSchemes {{ default linear; }}\nsnGradSchemes {{ default corrected; }}\nfvsc {{ default {stencil}; }}\n")
    with open(os.path.join(case_dir, "system", "fvSolution"), "w") as f:
        f.write(HDR.format(cls="dictionary", obj="fvSolution") + textwrap.dedent('''
            solvers
            {
                p { solver PCG; preconditioner DIC; tolerance 1e-12; relTol 0; maxIter 2000; }
                "(U|T)" { solver PBiCGStab; preconditioner DILU; tolerance 1e-12; relTol 0; maxIter 1500; }
            }
            '''))
    with open(os.path.join(case_dir, "system", "controlDict"), "w") as f:
        f.write(HDR.format(cls="dictionary", obj="controlDict") +
                "application QHDFoam;\nstartFrom startTime;\nstartTime 0;\nendTime 0.012;\ndeltaT 1e-3;\nwriteControl timeStep;\nwriteInterval 6;\n"
                "timePrecision 8;\n")
    return mesh


def hand_built_oracle(mesh, implicit, stencil="GaussVolPoint"):
    from qgdsolver_amd import qhdfoam
    from oracle import OracleQhdCase
    from util import oracle_mesh_of
    opt = qhdfoam.qhd_options(stencil=stencil, tauModel="HbyUQHD", aQGD=0.5, UQHD=0.8, rho0=1.2, mu=1.8e-2, Pr=0.71, beta=3.4e-3,
                              g=(0.0, -9.81, 0.0), deltaT=1e-3, pTol=1e-12, pRelTol=0.0, pMaxIter=2000, pRefCell=17, pRefValue=0.25,
                              implicitDiffusion=implicit, implicitTol=1e-14, implicitMaxIter=1500, precond=0)   # (the oracle's CG: tighter than the case's 1e-12)
    oc = OracleQhdCase(oracle_mesh_of(mesh), opt)
    for ip, name in enumerate(mesh.patch_names):
        T = ("fixedValue", 310.0) if name == "hot" else (("fixedValue", 290.0) if name == "cold" else ("zeroGradient", None))
        p = ("fixedGradient", 0.02) if name == "ceiling" else ("fixedGradient", 0.0)
        oc.set_bc(ip, U=("fixedValue", (0.0, 0.0, 0.0)), T=T, p=p)
    n = mesh.nCells
    oc.set_fields(np.zeros((n, 3)), np.full(n, 300.0), np.zeros(n))
    return oc


def test_read_qhd_case_setup(tmp_path):
    mesh = write_cavity_case(str(tmp_path))
    m2, opt, fields, bcs = ff.read_qhd_case_setup(str(tmp_path))
    assert m2.nCells == mesh.nCells and opt["stencil"] == "GaussVolPoint" and opt["deltaT"] == 1e-3
    assert opt["implicitDiffusion"] == 1                       # absent from QGD{}: the reference's default [QGDThermo.C L70-82]
    assert (opt["rho0"], opt["mu"], opt["Pr"], opt["beta"]) == (1.2, 1.8e-2, 0.71, 3.4e-3) and opt["g"] == (0.0, -9.81, 0.0)
    assert opt["tauModel"] == "HbyUQHD" and opt["UQHD"] == 0.8 and opt["aQGD"] == 0.5 and opt["pRefCell"] == 17 and opt["pRefValue"] == 0.25
    assert (opt["pTol"], opt["pRelTol"], opt["pMaxIter"]) == (1e-12, 0.0, 2000) and (opt["implicitTol"], opt["implicitMaxIter"]) == (1e-12, 1500)
    by = dict(zip(m2.patch_names, bcs))
    assert by["hot"]["T"] == ("fixedValue", 310.0) and by["cold"]["T"] == ("fixedValue", 290.0) and by["floor"]["T"] == ("zeroGradient", None)
    assert by["hot"]["U"][0] == "fixedValue" and by["ceiling"]["p"] == ("fixedGradient", 0.02)
    assert by["floor"]["p"] == ("fixedGradient", 0.0)          # qhdFlux inside QHDFoam = the gradient of its file [qhdFlux...C L166-168]
    assert np.all(fields["T"] == 300.0) and fields["U"].shape == (mesh.nCells, 3)
    # the explicit branch when the dictionary says so; the other closures' keys
    for k, (closure, want) in enumerate((("constTau", {"Tau": 2e-3}), ("T0byGr", {"T0": 1.5, "Gr": 400.0}), ("H2bynuQHD", {}))):
        d = tmp_path / f"c{k}"
        write_cavity_case(str(d), implicit_line="implicitDiffusion false;", closure=closure)
        _, o, _, _ = ff.read_qhd_case_setup(str(d))
        assert o["implicitDiffusion"] == 0 and o["tauModel"] == closure and all(o[a] == b for a, b in want.items()) and "implicitTol" not in o


def test_qhd_reader_refuses_what_the_path_does_not_do(tmp_path):
    write_cavity_case(str(tmp_path))
    tp = os.path.join(str(tmp_path), "constant", "thermophysicalProperties")
    text = open(tp).read()
    open(tp, "w").write(text.replace("QGDCoeffs HbyUQHD;", "QGDCoeffs constScPrModel1;"))
    with pytest.raises(ff.FoamFileError, match="closure"):
        ff.read_qhd_case_setup(str(tmp_path))
    open(tp, "w").write(text.replace("equationOfState rhoConst;", "equationOfState perfectGas;"))
    with pytest.raises(ff.FoamFileError, match="rhoConst"):
        ff.read_qhd_case_setup(str(tmp_path))
    open(tp, "w").write(text)
    cd = os.path.join(str(tmp_path), "system", "controlDict")
    open(cd, "a").write("adjustTimeStep yes;\n")
    with pytest.raises(ff.FoamFileError, match="adjustTimeStep"):
        ff.read_qhd_case_setup(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("implicit_line,implicit", [("", 1), ("implicitDiffusion false;", 0)])
def test_qhdfoam_application_round_trip(tmp_path, implicit_line, implicit):
    """the application on the written case: time directories at the write cadence, equal to the hand-built oracle, reading back to
    the device state bit for bit, restartable from the last one"""
    case_dir = str(tmp_path)
    mesh = write_cavity_case(case_dir, implicit_line=implicit_line)
    pr = subprocess.run([sys.executable, "-m", "qgdsolver_amd.QHDFoam", "-case", case_dir], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert pr.returncode == 0, pr.stderr[-1500:]
    out = pr.stdout
    assert f"implicitDiffusion {'true' if implicit else 'false'}" in out and "Time = 0.006" in out and "Time = 0.012" in out and out.rstrip().endswith("End")
    assert ("Ux:" in out) == bool(implicit) and "max/min of T" in out and "WARNING" not in out
    assert sorted(d for d in os.listdir(case_dir) if d[0].isdigit()) == ["0", "0.006", "0.012"]
    oc = hand_built_oracle(mesh, implicit)
    oc.step(12)
    m2 = ff.read_polymesh(os.path.join(case_dir, "constant", "polyMesh"))
    for name in ("U", "T", "p"):
        vals, patches = ff.read_field(os.path.join(case_dir, "0.012", name), m2)
        want = oc.field(name)
        err = np.abs(vals.reshape(want.shape) - want).max() / max(np.abs(want).max(), 1e-300)
        assert err <= 1e-9, (name, err)
    assert patches["ceiling"]["type"] == "fixedGradient" and patches["ceiling"]["gradient"][0, 0] == 0.02 and patches["hot"]["type"] == "fixedGradient"
    assert np.abs(oc.field("U")).max() > 1e-4                   # the cavity has started to turn
    # in process: load, step, write, read back bit for bit
    dev, gc = ff.load_qhd_case(case_dir, time="0.006")
    assert gc.options.implicitDiffusion == implicit and gc.options.pRefCell == 17
    gc.step(6)
    _, _, _, bcs = ff.read_qhd_case_setup(case_dir, "0.006")
    ff.write_qhd_time(gc, case_dir, "restart", bcs)
    for name in ("U", "T", "p"):
        vals, _ = ff.read_field(os.path.join(case_dir, "restart", name), gc.mesh)
        want = gc.field(name)
        assert np.array_equal(vals.reshape(want.shape), want), name
        # a restart from the written 0.006 lands on the uninterrupted run's 0.012 to the solver tolerances
        ref, _ = ff.read_field(os.path.join(case_dir, "0.012", name), gc.mesh)
        assert np.abs(vals - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300), name
    gc.close(); dev.close()
