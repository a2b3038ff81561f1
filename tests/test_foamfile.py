"""OpenFOAM ASCII polyMesh / field / dictionary reader: round trips and a hand-written case."""
import os
import textwrap

import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
from qgdsolver_amd import foamfile as ff
from util import make_mesh


@pytest.mark.parametrize("kind", ["box654_jitter", "box654_tri", "box654_poly", "plane2d", "step2d"])
def test_polymesh_round_trip_is_bit_exact(tmp_path, kind):
    mesh = make_mesh(kind)
    ff.write_polymesh(mesh, str(tmp_path / "polyMesh"))
    back = ff.read_polymesh(str(tmp_path / "polyMesh"))
    for name in ("points", "faceOffsets", "facePoints", "owner", "neighbour", "patchStart", "patchSize", "patchType",
                 "Sf", "Cf", "C", "V", "weights", "nonOrthDeltaCoeffs"):
        assert np.array_equal(mesh.array(name), back.array(name)), name
    assert back.nCells == mesh.nCells and back.nGeometricD == mesh.nGeometricD
    assert len(back.patch_names) == mesh.nPatches


def test_field_round_trip_and_uniform_entries(tmp_path):
    mesh = make_mesh("box654_jitter")
    mesh.patch_names = ["xmin", "xmax", "ymin", "ymax", "zmin", "zmax"]
    rng = np.random.default_rng(5)
    U = rng.standard_normal((mesh.nCells, 3))
    sizes = mesh.array("patchSize")
    patches = {pn: ("fixedValue", rng.standard_normal((int(sizes[i]), 3))) if i == 0 else ("zeroGradient", None)
               for i, pn in enumerate(mesh.patch_names)}
    path = str(tmp_path / "0" / "U")
    ff.write_field(path, mesh, "U", U, patches, "[0 1 -1 0 0 0 0]")
    Ui, bU = ff.read_field(path, mesh)
    assert np.array_equal(Ui, U)
    assert bU["xmin"]["type"] == "fixedValue" and np.array_equal(bU["xmin"]["value"], patches["xmin"][1])
    assert bU["zmax"]["type"] == "zeroGradient" and bU["zmax"]["value"] is None
    # uniform spellings
    ff.write_field(str(tmp_path / "0" / "T"), mesh, "T", np.float64(300.0), {pn: ("fixedValue", np.float64(1.5)) for pn in mesh.patch_names})
    Ti, bT = ff.read_field(str(tmp_path / "0" / "T"), mesh)
    assert Ti.shape == (mesh.nCells, 1) and np.all(Ti == 300.0) and np.all(bT["ymin"]["value"] == 1.5)


def test_dictionary_grammar():
    d = ff.parse_foam_text(textwrap.dedent('''
        /*--------------------------------*- C++ -*----------------------------------*\\
        | a banner with // slashes and (parens) |
        \\*---------------------------------------------------------------------------*/
        FoamFile { version 2.0; format ascii; class dictionary; object fvSchemes; }
        // line comment
        ddtSchemes { default Euler; }
        fvsc { default GaussVolPoint; }
        "(rho|rhoU)" { solver diagonal; }
        vec (1 2.5 -3e-2);
        inGroups 1(wall);
        dims [0 2 -2 0 0 0 0];
        nested { a { b { c 1; } } }
        str "hello world";
    '''))
    assert d["fvsc"]["default"] == "GaussVolPoint" and d["ddtSchemes"]["default"] == "Euler"
    assert d["(rho|rhoU)"]["solver"] == "diagonal"
    assert d["vec"] == [1, 2.5, -0.03] and d["inGroups"] == ["wall"]
    assert d["dims"] == ("dimensions", [0, 2, -2, 0, 0, 0, 0])
    assert d["nested"]["a"]["b"]["c"] == 1 and d["str"] == "hello world"
    with pytest.raises(ff.FoamFileError):
        ff.parse_foam_text("FoamFile { format binary; } a 1;")
    with pytest.raises(ff.FoamFileError):
        ff.parse_foam_text("a { b 1; ")
    with pytest.raises(ff.FoamFileError):
        ff.parse_foam_text("a 1 }")


def write_step_case(case_dir, stencil="leastSquares", nx=30, ny=10, schemes=None, symmetry_walls=False, field_types=None):
    """a small forwardStep-style case directory written the way a QGDFoam user would.  schemes: {sub-dictionary of fvSchemes: body}
    over the usual ones; symmetry_walls: bottom and top are symmetryPlane patches as in OpenFOAM's forwardStep tutorial;
    field_types: {(field, patch): entry body} over the usual boundary conditions"""
    mesh = q.PolyMesh.forward_step(nx, ny, nx // 5, ny // 5)
    if symmetry_walls:
        ptypes = mesh.array("patchType").copy()
        ptypes[2] = ptypes[3] = L.PATCH_SYMMETRYPLANE
        mesh = q.PolyMesh.from_arrays(mesh.array("points"), mesh.array("faceOffsets"), mesh.array("facePoints"), mesh.array("owner"),
                                      mesh.array("neighbour"), mesh.nCells, mesh.array("patchStart"), mesh.array("patchSize"), ptypes)
    mesh.patch_names = ["inlet", "outlet", "bottom", "top", "obstacle", "frontAndBack"][:mesh.nPatches]
    ff.write_polymesh(mesh, os.path.join(case_dir, "constant", "polyMesh"))
    names = mesh.patch_names
    pt = mesh.array("patchType")

    def bf(entries, field):
        out = []
        for i, n in enumerate(names):
            body = "type empty;" if pt[i] == L.PATCH_EMPTY else ("type symmetryPlane;" if pt[i] == L.PATCH_SYMMETRYPLANE
                                                                 else entries.get(n, entries["default"]))
            body = (field_types or {}).get((field, n), body)
            out.append(f"    {n} {{ {body} }}")
        return "boundaryField\n{\n" + "\n".join(out) + "\n}\n"

    hdr = "FoamFile {{ version 2.0; format ascii; class {cls}; object {obj}; }}\n"
    os.makedirs(os.path.join(case_dir, "0"))
    os.makedirs(os.path.join(case_dir, "system"))
    with open(os.path.join(case_dir, "0", "U"), "w") as f:
        f.write(hdr.format(cls="volVectorField", obj="U") + "dimensions [0 1 -1 0 0 0 0];\ninternalField uniform (3 0 0);\n" +
                bf({"inlet": "type fixedValue; value uniform (3 0 0);", "outlet": "type zeroGradient;", "default": "type slip;"}, "U"))
    with open(os.path.join(case_dir, "0", "T"), "w") as f:
        f.write(hdr.format(cls="volScalarField", obj="T") + "dimensions [0 0 0 1 0 0 0];\ninternalField uniform 1;\n" +
                bf({"inlet": "type fixedValue; value uniform 1;", "default": "type zeroGradient;"}, "T"))
    with open(os.path.join(case_dir, "0", "p"), "w") as f:
        f.write(hdr.format(cls="volScalarField", obj="p") + "dimensions [1 -1 -2 0 0 0 0];\ninternalField uniform 1;\n" +
                bf({"inlet": "type fixedValue; value uniform 1;", "outlet": "type zeroGradient;", "default": "type qgdFlux;"}, "p"))
    with open(os.path.join(case_dir, "constant", "thermophysicalProperties"), "w") as f:
        f.write(hdr.format(cls="dictionary", obj="thermophysicalProperties") + textwrap.dedent(f'''
            thermoType {{ type hePsiQGDThermo; mixture pureMixture; transport const; thermo eConst;
                         equationOfState perfectGas; specie specie; energy sensibleInternalEnergy; }}
            mixture
            {{
                specie {{ molWeight {ff.RR * 1.4!r}; }}   // R = 1/1.4: c = 1 at T = 1
                thermodynamics {{ Cv {1.0 / 1.4 / 0.4!r}; Hf 0; }}
                transport {{ mu 0; Pr 1; }}
            }}
            QGD
            {{
                implicitDiffusion false;
                QGDCoeffs constScPrModel1;
                constScPrModel1Dict {{ ScQGD 1; PrQGD 1; }}
            }}
            '''))
    with open(os.path.join(case_dir, "system", "fvSchemes"), "w") as f:
        sections = {"ddtSchemes": "default Euler;", "gradSchemes": "default Gauss linear;", "divSchemes": "default none;",
                    "laplacianSchemes": "default Gauss linear corrected;", "interpolationSchemes": "default linear;",
                    "snGradSchemes": "default corrected;", "fvsc": f"default {stencil};"}
        sections.update(schemes or {})
        f.write(hdr.format(cls="dictionary", obj="fvSchemes") +
                "".join(f"{k} {{ {v} }}\n" for k, v in sections.items() if v is not None))
    with open(os.path.join(case_dir, "system", "controlDict"), "w") as f:
        f.write(hdr.format(cls="dictionary", obj="controlDict") +
                "application QGDFoam;\nstartTime 0;\nendTime 1;\ndeltaT 5e-4;\nadjustTimeStep no;\nmaxCo 0.2;\n")
    return mesh


def test_read_case_setup(tmp_path):
    mesh = write_step_case(str(tmp_path))
    m2, opt, fields, bcs = ff.read_case_setup(str(tmp_path))
    assert m2.nCells == mesh.nCells and m2.nGeometricD == 2
    assert opt["stencil"] == "leastSquares" and opt["deltaT"] == 5e-4 and opt["adjustTimeStep"] == 0
    assert opt["implicitDiffusion"] == 0 and opt["ScQGD"] == 1.0 and opt["PrQGD"] == 1.0 and opt["alphaQGD"] == 0.5
    assert abs(opt["R"] - 1 / 1.4) < 1e-15 and abs(opt["Cv"] - 1 / 1.4 / 0.4) < 1e-15 and opt["maxCo"] == 0.2
    assert fields["U"].shape == (mesh.nCells, 3) and np.all(fields["U"][:, 0] == 3) and np.all(fields["T"] == 1)
    by = dict(zip(m2.patch_names, bcs))
    assert by["inlet"]["U"][0] == "fixedValue" and list(by["inlet"]["U"][1]) == [3, 0, 0] and by["inlet"]["p"] == ("fixedValue", 1.0)
    assert by["outlet"]["U"] == ("zeroGradient", None)
    walls = [n for n in m2.patch_names if n not in ("inlet", "outlet") and by[n]["U"][0] != "none"]
    assert walls and all(by[n]["U"][0] == "slip" and by[n]["p"][0] == "qgdFlux" and by[n]["T"][0] == "zeroGradient" for n in walls)
    empties = [n for n in m2.patch_names if by[n]["U"][0] == "none"]
    assert empties and all(by[n]["p"][0] == "none" for n in empties)
    # the reference's default when the entry is absent is implicitDiffusion true [QGDThermo.C L70-82]
    tp = os.path.join(str(tmp_path), "constant", "thermophysicalProperties")
    text = open(tp).read().replace("implicitDiffusion false;", "")
    open(tp, "w").write(text)
    assert ff.read_case_setup(str(tmp_path))[1]["implicitDiffusion"] == 1
    # alphaQGD is READ_IF_PRESENT from the time directory [QGDCoeffs.C L119-160]
    ff.write_field(os.path.join(str(tmp_path), "0", "alphaQGD"), m2, "alphaQGD", np.float64(0.3),
                   {n: (("empty", None) if by[n]["U"][0] == "none" else ("zeroGradient", None)) for n in m2.patch_names})
    assert ff.read_case_setup(str(tmp_path))[1]["alphaQGD"] == 0.3


def test_unsupported_entries_fail_loudly(tmp_path):
    write_step_case(str(tmp_path))
    bpath = os.path.join(str(tmp_path), "constant", "polyMesh", "boundary")
    text = open(bpath).read()
    open(bpath, "w").write(text.replace("type            patch;", "type            processor;", 1))
    with pytest.raises(ff.FoamFileError, match="processor"):
        ff.read_polymesh(os.path.dirname(bpath))
    open(bpath, "w").write(text)
    upath = os.path.join(str(tmp_path), "0", "U")
    utext = open(upath).read()
    open(upath, "w").write(utext.replace("type zeroGradient;", "type waveTransmissive;"))
    with pytest.raises(ff.FoamFileError, match="waveTransmissive"):
        ff.read_case_setup(str(tmp_path))


def test_time_names_follow_openfoam():
    from qgdsolver_amd.QGDFoam import time_name
    assert time_name(0.0) == "0" and time_name(0.005) == "0.005" and time_name(0.0125) == "0.0125"
    assert time_name(1.0) == "1" and time_name(1e-7) == "1e-07" and time_name(0.1 + 0.2) == "0.3"
    assert time_name(123456.789, 8) == "123456.79"


def test_gzipped_case_files(tmp_path):
    """writeCompression on: every file of the case may be <name>.gz"""
    import gzip
    import shutil

    write_step_case(str(tmp_path))
    ref = ff.read_case_setup(str(tmp_path))
    for sub in ("constant/polyMesh/points", "constant/polyMesh/faces", "constant/polyMesh/owner", "constant/polyMesh/neighbour",
                "constant/polyMesh/boundary", "0/U", "0/T", "0/p"):
        path = os.path.join(str(tmp_path), sub)
        with open(path, "rb") as fi, gzip.open(path + ".gz", "wb") as fo:
            shutil.copyfileobj(fi, fo)
        os.remove(path)
    got = ff.read_case_setup(str(tmp_path))
    assert np.array_equal(got[0].array("points"), ref[0].array("points")) and np.array_equal(got[0].array("facePoints"), ref[0].array("facePoints"))
    assert got[1] == ref[1] and all(np.array_equal(got[2][k], ref[2][k]) for k in ref[2])


def test_time_controls_defaults_and_cTau(tmp_path):
    """setDeltaT-QGDQHD.H L45: cTau from controlDict (default 0.75); createTimeControls defaults (L0): maxCo 1, maxDeltaT GREAT"""
    write_step_case(str(tmp_path))
    cpath = os.path.join(str(tmp_path), "system", "controlDict")
    text = open(cpath).read()
    opt = ff.read_case_setup(str(tmp_path))[1]
    assert opt["maxCo"] == 0.2 and opt["cTau"] == 0.75 and opt["maxDeltaT"] >= 1e299
    open(cpath, "w").write(text.replace("maxCo 0.2;", "adjustTimeStep yes;\ncTau 0.3;\nmaxDeltaT 2e-3;"))
    opt = ff.read_case_setup(str(tmp_path))[1]
    assert opt["maxCo"] == 1.0 and opt["cTau"] == 0.3 and opt["maxDeltaT"] == 2e-3
    o = q.default_options(**opt)
    assert (o.maxCo, o.cTau, o.maxDeltaT) == (1.0, 0.3, 2e-3)


def test_nonuniform_alphaQGD_and_ScQGD_files(tmp_path):
    """alphaQGD / ScQGD are volScalarFields read if present [QGDCoeffs.C L119-160, constScPrModel1.C L66-79]"""
    mesh = write_step_case(str(tmp_path))
    m2, opt, fields, bcs = ff.read_case_setup(str(tmp_path))
    assert "alphaQGD" not in fields and "ScQGD" not in fields
    rng = np.random.default_rng(3)
    a = 0.3 + 0.4 * rng.random(m2.nCells)
    by = dict(zip(m2.patch_names, bcs))
    sizes = dict(zip(m2.patch_names, m2.array("patchSize")))
    patches = {n: (("empty", None) if by[n]["U"][0] == "none" else
                   (("calculated", np.full(int(sizes[n]), 0.45)) if n == "inlet" else ("zeroGradient", None))) for n in m2.patch_names}
    ff.write_field(os.path.join(str(tmp_path), "0", "alphaQGD"), m2, "alphaQGD", a, patches)
    ff.write_field(os.path.join(str(tmp_path), "0", "ScQGD"), m2, "ScQGD", np.float64(0.8), {n: patches[n] if patches[n][0] == "empty" else ("zeroGradient", None) for n in m2.patch_names})
    m3, opt3, fields3, _ = ff.read_case_setup(str(tmp_path))
    assert opt3["ScQGD"] == 0.8 and "ScQGD" not in fields3          # uniform file -> the scalar option
    cells, bnd = fields3["alphaQGD"]
    assert np.array_equal(cells, a) and bnd.size == m3.nBoundaryFaces
    nif, own = m3.nInternalFaces, m3.array("owner")
    ps, pz = m3.array("patchStart"), m3.array("patchSize")
    for i, n in enumerate(m3.patch_names):
        sl = slice(int(ps[i]) - nif, int(ps[i]) - nif + int(pz[i]))
        if n == "inlet":
            assert np.all(bnd[sl] == 0.45)
        elif patches[n][0] == "zeroGradient":
            assert np.array_equal(bnd[sl], a[own[int(ps[i]): int(ps[i]) + int(pz[i])]])


def test_list_head_is_not_found_inside_numbers_or_words():
    d = ff.parse_foam_text("FoamFile { format ascii; }\nv 0.125 (1 0 0);\nname3 (4 5);\nw 2(7 8);")
    assert d["w"] == [7, 8]
    assert d["v"][0] == 0.125
    # a non-numeric list of 64 or more items goes to the token parser, not to numpy
    words = " ".join(f"p{i}" for i in range(70))
    d = ff.parse_foam_text(f"FoamFile {{ format ascii; }}\nnames 70({words});")
    assert d["names"][0] == "p0" and len(d["names"]) == 70


# ---------------------------------------------------------------------------------------------------------------------
# fvSchemes sub-dictionaries the face-flux path consults [QGDInterpolate.H L42-104, fvsc.C L51-58] and constraint patches
# ---------------------------------------------------------------------------------------------------------------------
def test_symmetry_plane_walls_read_as_the_constraint_they_are(tmp_path):
    """OpenFOAM's forwardStep tutorial: bottom and top are symmetryPlane patches; the field entries must carry that type and the
    fields ARE the constraint (U reflected, scalars zero-gradient), whatever the solver-level boundary conditions elsewhere"""
    write_step_case(str(tmp_path), symmetry_walls=True)
    m2, opt, fields, bcs = ff.read_case_setup(str(tmp_path))
    by = dict(zip(m2.patch_names, bcs))
    for wall in ("bottom", "top"):
        assert by[wall] == {"U": ("slip", None), "T": ("zeroGradient", None), "p": ("zeroGradient", None)}
    assert by["obstacle"]["U"][0] == "slip" and by["obstacle"]["p"][0] == "qgdFlux"
    # a field entry of another type on a constraint patch is OpenFOAM's "inconsistent patch and patchField types"
    write_step_case(str(tmp_path / "bad"), symmetry_walls=True, field_types={("U", "top"): "type zeroGradient;"})
    with pytest.raises(ff.FoamFileError, match="inconsistent patch and patchField types.*symmetryPlane"):
        ff.read_case_setup(str(tmp_path / "bad"))
    # ... and a constraint type on an ordinary patch needs a patch of that type
    write_step_case(str(tmp_path / "bad2"), field_types={("p", "top"): "type symmetryPlane;"})
    with pytest.raises(ff.FoamFileError, match="needs a patch of that type"):
        ff.read_case_setup(str(tmp_path / "bad2"))


@pytest.mark.parametrize("word", ["cyclic", "wedge"])
def test_cyclic_and_wedge_cases_are_refused(tmp_path, word):
    write_step_case(str(tmp_path))
    bpath = os.path.join(str(tmp_path), "constant", "polyMesh", "boundary")
    text = open(bpath).read()
    open(bpath, "w").write(text.replace("type            patch;", f"type            {word};", 1))
    with pytest.raises(ff.FoamFileError, match=f"{word} patch"):
        ff.read_case_setup(str(tmp_path))


def test_schemes_that_change_nothing_are_accepted(tmp_path):
    """`linear` goes through fvc::interpolate with the same weights; `Gauss linear` through fvc::flux with the same numbers; a per-term
    fvsc entry that repeats the default is the default"""
    write_step_case(str(tmp_path), schemes={
        "interpolationSchemes": "default linear; interpolate(rho) linear; interpolate(U) linear;",
        "divSchemes": "default none; div(phiJm,U) Gauss linear; div(phi,K) Gauss limitedLinear 1;",   # the last is not one of qgdFlux's names
        "fvsc": "default leastSquares; grad(p) leastSquares;"})
    opt = ff.read_case_setup(str(tmp_path))[1]
    assert opt["stencil"] == "leastSquares" and opt["fluxSchemeU"] == 0 and opt["fluxSchemeH"] == 0
    write_step_case(str(tmp_path / "none"), schemes={"interpolationSchemes": "default none;"})
    assert ff.read_case_setup(str(tmp_path / "none"))[1]["stencil"] == "leastSquares"


def test_per_term_fvsc_entries_become_options(tmp_path):
    """fvsc{default leastSquares; grad(p) reduced;} [fvsc.C L51-58]: the term's own word travels as qgd_case_options::termStencil"""
    write_step_case(str(tmp_path), schemes={"fvsc": "default leastSquares; grad(p) reduced; grad(U) leastSquares;"})
    opt = ff.read_case_setup(str(tmp_path))[1]
    assert opt["stencil"] == "leastSquares" and opt["termStencils"] == {"grad(p)": "reduced"}
    o = q.default_options(**opt)
    assert list(o.termStencil) == [0, 0, 0, 1 + L.FVSC_REDUCED] and o.stencil == L.FVSC_LEASTSQUARES
    # no default: every term carries its own word
    write_step_case(str(tmp_path / "nodefault"), schemes={"fvsc": "grad(U) GaussVolPoint; grad(e) GaussVolPoint; grad(rho) GaussVolPoint; grad(p) reduced;"})
    opt = ff.read_case_setup(str(tmp_path / "nodefault"))[1]
    assert opt["stencil"] == "GaussVolPoint" and opt["termStencils"] == {"grad(p)": "reduced"}


def test_gauss_upwind_fluxes_become_options(tmp_path):
    write_step_case(str(tmp_path), schemes={"divSchemes": "default none; div(phiJm,U) Gauss upwind;"})
    opt = ff.read_case_setup(str(tmp_path))[1]
    assert (opt["fluxSchemeU"], opt["fluxSchemeH"]) == (1, 0)
    o = q.default_options(**opt)
    assert (o.fluxSchemeU, o.fluxSchemeH) == (L.FLUX_UPWIND, L.FLUX_LINEAR)
    # a quoted (regular-expression) key matches like dictionary::found does
    write_step_case(str(tmp_path / "re"), schemes={"divSchemes": 'default none; "div\\(phiJm,.*\\)" Gauss upwind;'})
    opt = ff.read_case_setup(str(tmp_path / "re"))[1]
    assert (opt["fluxSchemeU"], opt["fluxSchemeH"]) == (1, 1)


@pytest.mark.parametrize("schemes,needle", [
    ({"interpolationSchemes": "default cubic;"}, "interpolationSchemes.default 'cubic'"),
    ({"interpolationSchemes": "default linear; interpolate(rhoU) vanLeer;"}, r"interpolationSchemes.interpolate\(rhoU\) 'vanLeer'"),
    ({"divSchemes": "default none; div(phiJm,H) Gauss limitedLinear 1;"}, r"divSchemes.div\(phiJm,H\) 'Gauss limitedLinear 1'"),
    ({"fvsc": "default leastSquares; grad(p) reduced; grad(e) GaussVolPoint;"}, "more than two distinct stencils"),
    ({"fvsc": "grad(U) reduced;"}, "neither 'grad\\(e\\)' nor 'default'"),
    ({"interpolationSchemes": None}, "'interpolationSchemes' is missing"),
    ({"divSchemes": None}, "'divSchemes' is missing"),
    ({"ddtSchemes": "default backward;"}, "ddtSchemes.default 'backward'"),
])
def test_schemes_the_path_does_not_compute_are_refused_by_name(tmp_path, schemes, needle):
    write_step_case(str(tmp_path), schemes=schemes)
    with pytest.raises(ff.FoamFileError, match=needle):
        ff.read_case_setup(str(tmp_path))


def test_laplacian_and_grad_schemes_of_the_implicit_branch(tmp_path):
    """implicitDiffusion reads laplacianSchemes (fvm::laplacian) and gradSchemes (fvc::grad(U)): Gauss linear [un]corrected, the
    corrected form only where it is the same operator (an orthogonal mesh)"""
    def with_implicit(case_dir, **kw):
        write_step_case(case_dir, **kw)
        tp = os.path.join(case_dir, "constant", "thermophysicalProperties")
        text = open(tp).read().replace("implicitDiffusion false;", "implicitDiffusion true;")
        open(tp, "w").write(text)
    with_implicit(str(tmp_path / "a"))
    assert ff.read_case_setup(str(tmp_path / "a"))[1]["implicitDiffusion"] == 1     # Gauss linear corrected on the orthogonal step mesh
    with_implicit(str(tmp_path / "b"), schemes={"laplacianSchemes": "default Gauss linear limited 0.5;"})
    with pytest.raises(ff.FoamFileError, match="laplacianSchemes.default 'Gauss linear limited 0.5'"):
        ff.read_case_setup(str(tmp_path / "b"))
    with_implicit(str(tmp_path / "c"), schemes={"gradSchemes": "default leastSquares;"})
    with pytest.raises(ff.FoamFileError, match="gradSchemes.default 'leastSquares'"):
        ff.read_case_setup(str(tmp_path / "c"))
    # the explicit branch reads neither
    write_step_case(str(tmp_path / "d"), schemes={"gradSchemes": "default leastSquares;"})
    assert ff.read_case_setup(str(tmp_path / "d"))[1]["implicitDiffusion"] == 0
