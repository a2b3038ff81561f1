"""The oracle against numbers evaluated mechanically from the reference's own listing text (tests/golden/make_ref_expr.py:
the C++ expressions of the listings executed with an emulation of OpenFOAM's vector/tensor operators).  Every golden
configuration is one internal face between cells with prescribed centres, so the oracle's PUBLIC operators are what is
checked -- stencil coefficients, slot order of the dfdxif macro, the tensor layout, the interior-triangle pattern, the
leastSquares weights and degeneracy rule, and the whole updateFields.H / updateFluxes.H flux algebra."""
import numpy as np
import pytest

from oracle import OracleCase, OracleMesh
import oracle
import ref_expr_cases as rc

import qgdsolver_amd as q

TOL = 2e-13


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def oracle_mesh(prim, geom):
    om = OracleMesh(prim)
    om.set_geometry(geom["Sf"], geom["Cf"], geom["C"], geom["V"])
    return om


def test_gaussvolpoint_3d_coefficients_and_macro():
    g = rc.load("gvp3d")
    assert set(g["nv"]) == {3, 4}
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        assert om.info()["nGeometricD"] == 3
        rc_s, gs = om.fvsc("GaussVolPoint", "grad_s", g["cell_s"][i], np.zeros(0))
        rc_v, gv = om.fvsc("GaussVolPoint", "grad_v", g["cell_v"][i], np.zeros(0))
        assert rc_s == 0 and rc_v == 0
        assert rel(gs[0], g["grad_s"][i]) <= TOL, (i, nv, gs[0], g["grad_s"][i])
        assert rel(gv[0], g["grad_v"][i]) <= TOL, (i, nv)
        if nv == 3:   # the listing's interior-triangle pattern: every row holds (dxUx, dyUy, dzUz)
            assert np.array_equal(g["grad_v"][i][0:3], g["grad_v"][i][3:6]) and np.array_equal(g["grad_v"][i][0:3], g["grad_v"][i][6:9])
        # the divergences [calcDivfIF L552-560 (vector), L633-659 (tensor: d_i T_ij, component 3*i + j)]
        rc_dv, dv = om.fvsc("GaussVolPoint", "div_v", g["cell_v"][i], np.zeros(0))
        rc_dt, dt = om.fvsc("GaussVolPoint", "div_t", g["cell_t"][i], np.zeros(0))
        assert rc_dv == 0 and rc_dt == 0
        assert rel(dv[0], g["div_v"][i]) <= TOL, (i, nv, dv[0], g["div_v"][i])
        assert rel(dt[0], g["div_t"][i]) <= TOL, (i, nv, dt[0], g["div_t"][i])
        om.close()


def bnd_ops(g, i):
    """(op, cell values of the two cells, patch value, expected) of boundary golden configuration i; cell 1 is a bystander"""
    return (("grad_s", [g["cell_s"][i], 0.3], [g["bnd_s"][i]], g["grad_s"][i]),
            ("grad_v", [g["cell_v"][i], [0.1, 0.2, 0.3]], [g["bnd_v"][i]], g["grad_v"][i]),
            ("div_v", [g["cell_v"][i], [0.1, 0.2, 0.3]], [g["bnd_v"][i]], g["div_v"][i]),
            ("div_t", [g["cell_t"][i], np.arange(9.0)], [g["bnd_t"][i]], g["div_t"][i]))


def test_gaussvolpoint_3d_boundary_faces():
    """the boundary-face text of GaussVolPointBase3D.C: mirror point, coefficients with owner / mirror slots, psin = patch value +
    snGrad bmvON/2, macro dfdxbf, for scalar and vector gradients and vector and tensor divergences, quads and triangles"""
    g = rc.load("gvp3d_bnd")
    assert set(g["nv"]) == {3, 4}
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.boundary_face_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        assert om.info()["nGeometricD"] == 3
        for op, cell, bnd, want in bnd_ops(g, i):
            st, got = om.fvsc("GaussVolPoint", op, np.array(cell, float), np.array(bnd, float))
            assert st == 0
            assert rel(got[1], want) <= TOL, (i, nv, op, got[1], want)
        om.close()


def lsq_bnd_mesh(g, i):
    from qgdsolver_amd import _lib as L
    ie3 = int(g["ie3"][i])
    return rc.boundary_face_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], back_axis=(ie3 + 1) % 3, empty_normals=[rc.unit(ie3)],
                                 patch_type=L.PATCH_SYMMETRYPLANE if g["symmetry"][i] else L.PATCH_GENERIC)


def test_leastsquares_boundary_faces():
    """extendedFaceStencilScalarGrad.C L86-109: nf * snGrad on ordinary patches, zero on symmetryPlane (constraint) patches"""
    g = rc.load("lsq_bnd")
    assert set(g["symmetry"]) == {0, 1}
    for i in range(len(g["ie3"])):
        om = oracle_mesh(*lsq_bnd_mesh(g, i))
        st, gs = om.fvsc("leastSquares", "grad_s", np.array([g["f"][i], 0.3]), np.array([g["fb"][i], 0.0]))
        assert st == 0
        if g["symmetry"][i]:
            assert np.array_equal(gs[1], np.zeros(3)) and np.array_equal(g["grad"][i], np.zeros(3))
        else:
            assert rel(gs[1], g["grad"][i]) <= TOL, (i, gs[1], g["grad"][i])
        om.close()


def test_reduced_stencil_operators():
    """reducedFaceNormalStencil.C L71-105: nf * snGrad (outer product, layout d_i psi_j) and nf & snGrad for the four operators"""
    g = rc.load("reduced")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        for op, cell in (("grad_s", g["cell_s"][i]), ("grad_v", g["cell_v"][i]), ("div_v", g["cell_v"][i]), ("div_t", g["cell_t"][i])):
            st, got = om.fvsc("reduced", op, cell, np.zeros(0))
            assert st == 0
            assert rel(got[0], g[op][i]) <= TOL, (i, nv, op, got[0], g[op][i])
        om.close()


def test_gaussvolpoint_2d_boundary_faces():
    """GaussVolPointBase2D.C boundary faces: v42 = 2 (Cf - C), the two vertices above the cell centre, c1e..c4e, psi2 = patch value +
    snGrad |v42|/2, the apply of L352-359"""
    g = rc.load("gvp2d_bnd")
    for i in range(len(g["ie3"])):
        ie3 = int(g["ie3"][i])
        prim, geom = rc.boundary_face_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], back_axis=(ie3 + 1) % 3, empty_normals=[rc.unit(ie3)])
        om = oracle_mesh(prim, geom)
        info = om.info()
        assert info["nGeometricD"] == 2 and info["geometricD"][ie3] == -1
        st, gs = om.fvsc("GaussVolPoint", "grad_s", np.array([g["f"][i], 0.3]), np.array([g["fb"][i], 0.0]))
        assert st == 0
        assert rel(gs[1], g["grad"][i]) <= TOL, (i, ie3, gs[1], g["grad"][i])
        om.close()


def test_gaussvolpoint_2d_coefficients_and_apply():
    g = rc.load("gvp2d")
    for i in range(len(g["ie3"])):
        ie3 = int(g["ie3"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], empty_normals=[rc.unit(ie3)])
        om = oracle_mesh(prim, geom)
        info = om.info()
        assert info["nGeometricD"] == 2 and info["geometricD"][ie3] == -1
        st, gs = om.fvsc("GaussVolPoint", "grad_s", g["f"][i], np.zeros(1))
        assert st == 0
        assert rel(gs[0], g["grad"][i]) <= TOL, (i, ie3, gs[0], g["grad"][i])
        om.close()


def test_leastsquares_weights_degeneracy_and_apply():
    g = rc.load("lsq")
    assert 0 < g["deg"].sum() < len(g["deg"])     # both branches of det(G) < 1 occur
    for i in range(len(g["n"])):
        n = int(g["n"][i])
        prim, geom = rc.lsq_mesh(n, g["Cf"][i], g["centres"][i], one_d=(n == 2))
        om = oracle_mesh(prim, geom)
        assert om.info()["nGeometricD"] == (1 if n == 2 else 2)
        nb = prim["owner"].size - 1
        cell = g["iF"][i][:n]
        bnd = np.zeros(nb)
        st, got = om.fvsc("leastSquares", "grad_s", cell, bnd)
        assert st == 0
        if g["deg"][i]:
            st2, red = om.fvsc("reduced", "grad_s", cell, bnd)   # degenerate faces fall back to nf*snGrad [ScalarGrad.C L76-83]
            assert np.array_equal(got[0], red[0]), i
        else:
            assert rel(got[0], g["grad"][i]) <= TOL, (i, n, got[0], g["grad"][i])
        om.close()


FACE_FIELDS = ("rhof", "Uf", "pf", "cf", "Hf", "alphauf", "muf", "tauQGDf", "hQGDf", "gradUf", "gradef", "gradRhof", "gradPf",
               "phiwStar", "phiJm", "phi", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU")


STEP_FIELDS = ["rho", "U", "e", "rhoU", "rhoE"]
THERMO_FIELDS = ["T", "p", "c", "psi"]


def case_options(g, i):
    return q.default_options(stencil="GaussVolPoint", R=float(g["R"][i]), Cv=float(g["Cv"][i]), mu=float(g["mu"][i]), Pr=float(g["Pr"][i]),
                             ScQGD=float(g["ScQGD"][i]), PrQGD=float(g["PrQGD"][i]), alphaQGD=float(g["alphaQGD"][i]), deltaT=float(g["deltaT"][i]))


def test_flux_assembly_of_one_face():
    g = rc.load("case2cell")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        assert rel(om.array("weights")[0], g["w"][i]) <= 1e-15
        oc = OracleCase(om, case_options(g, i))
        oc.set_fields(g["U"][i], g["T"][i], g["p"][i])
        oc.updateFluxes()
        for f in FACE_FIELDS:
            assert rel(oc.field(f)[0], g[f][i]) <= TOL, (i, nv, f, oc.field(f)[0], g[f][i])
        for f in ("muQGD", "alphauQGD", "tauQGD", "hQGD"):
            assert rel(oc.field(f), g[f][i]) <= TOL, (i, f)
        # ... and one explicit step: QGDRhoEqn.H L40-47, QGDUEqn.H L36-89, QGDEEqn.H L37-76 executed from the listing text (Euler ddt and
        # fvc::div = surfaceIntegrate emulated, L0) with the fluxes above
        oc.step(1)
        for f in STEP_FIELDS:
            assert rel(oc.field(f), g[f + "1"][i]) <= TOL, (i, nv, f, oc.field(f), g[f + "1"][i])
        # ... then thermo.correct() and p = rho / psi: hePsiQGDThermo.C L48-64, L123-124 and QGDFoam.C L152-154 from the text
        th = rc.load("thermo2cell")
        for f in THERMO_FIELDS:
            assert rel(oc.field(f), th[f + "1"][i]) <= TOL, (i, nv, f, oc.field(f), th[f + "1"][i])
        oc.close(); om.close()


def test_implicit_branch_face_expressions():
    """updateFluxes.H with implicitDiffusion true, from the listing text: Pif / qf without their Navier-Stokes / Fourier parts (L95-106,
    L131-135) and tauMC = qgdInterpolate(muEff dev2(T(fvc::grad(U)))), phiTauMC = Sf & tauMC (L107-111)"""
    g = rc.load("case2cell")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        opt = case_options(g, i)
        opt.implicitDiffusion = 1
        oc = OracleCase(om, opt)
        oc.set_fields(g["U"][i], g["T"][i], g["p"][i])
        oc.updateFluxes()
        for f, want in (("phiPi", "phiPi_impl"), ("phiQ", "phiQ_impl"), ("tauMC", "tauMC"), ("phiTauMC", "phiTauMC")):
            assert rel(oc.field(f)[0], g[want][i]) <= TOL, (i, nv, f, oc.field(f)[0], g[want][i])
        assert rel(oc.field("phiJm")[0], g["phiJm"][i]) <= TOL          # the mass flux does not depend on the branch
        oc.close(); om.close()


QHD_FIELDS = ["gradUf", "gradTf", "gradPf", "phiu", "phiwo", "taubyrhof", "Wf", "phiUf", "phiTf", "phiTauTReg"]


def qhd_inputs(g, i):
    """arguments of updateFluxes for golden configuration i (no boundary faces on the two-cell mesh)"""
    return dict(U=(g["U"][i], np.zeros((0, 3))), T=(g["T"][i], np.zeros(0)), rho=(g["rho"][i], np.zeros(0)), tauQGDf=np.array([g["tauQGDf"][i]]),
                beta=float(g["beta"][i]), g=g["g"][i], p=(g["p"][i], np.zeros(0)), phi=np.array([g["phi"][i]]))


def test_qhd_face_expressions():
    """QHDFoam/updateFields.H L36-73, updateFluxes.H L33-38, QHDUEqn.H L36-43, QHDTEqn.H L65-66 evaluated from the listing text
    against the oracle's qhd_fluxes: gradients, phiu, phiwo (body force included), taubyrhof, Wf, phiUf = phi Uf - Sf.(Uf Wf), phiTf,
    phiTauTReg"""
    g = rc.load("qhdface")
    assert set(g["nv"]) == {3, 4}
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        assert rel(om.array("weights")[0], g["w"][i]) <= 1e-15
        a = qhd_inputs(g, i)
        res = oracle.qhd_fluxes(om, "GaussVolPoint", a["U"], a["T"], a["rho"], a["tauQGDf"], a["beta"], a["g"], p=a["p"], phi=a["phi"])
        assert sorted(res) == sorted(QHD_FIELDS)
        for f in QHD_FIELDS:
            assert rel(res[f][0], g[f][i]) <= TOL, (i, nv, f, res[f][0], g[f][i])
        om.close()


def test_species_flux_expressions():
    """reactingLagrangianQGDFoam/updateFluxes.H L122-127 (+ updateFields.H L38) from the listing text against the oracle's species block"""
    from qgdsolver_amd import qgdfoam
    g = rc.load("species")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        prim, geom = rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)

        class FakeDev:
            mesh = om
        res = qgdfoam.speciesFlux(FakeDev, "GaussVolPoint", (g["Y"][i], np.zeros(0)), (g["U"][i], np.zeros((0, 3))), [g["phiJm"][i]], [g["phi"][i]],
                                  [g["tauQGDf"][i]], call=lambda sch, *a: oracle.species_flux(om, sch, *a))
        for f in ("gradYf", "phiJmY", "diffusiveFlux"):
            assert rel(res[f][0], g[f][i]) <= TOL, (i, nv, f, res[f][0], g[f][i])
        om.close()


# ---- round 3: the sections that were still outside the mechanical pin ------------------------------------------------------
def test_gaussvolpoint_2d_vector_gradient_and_divergences():
    """GaussVolPointBase.C L90-100 (the re-pack of the three per-component 2-D gradients: component 3*i + j = d_i U_j) and the 2-D
    divergences of GaussVolPointBase2D.C L386-396 (vector), L447-450 + L464 ff. (tensor)"""
    g = rc.load("gvp2d_vec")
    for i in range(len(g["ie3"])):
        ie3 = int(g["ie3"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], empty_normals=[rc.unit(ie3)]))
        assert om.info()["nGeometricD"] == 2
        for op, cell, nb, key in (("grad_v", g["U"][i], 3, "grad_v"), ("div_v", g["U"][i], 3, "div_v"), ("div_t", g["Tn"][i], 9, "div_t")):
            st, got = om.fvsc("GaussVolPoint", op, cell, np.zeros((1, nb)))
            assert st == 0
            assert rel(got[0], g[key][i]) <= TOL, (i, ie3, op, got[0], g[key][i])
        om.close()


def test_gaussvolpoint_3d_faces_with_more_than_four_vertices():
    """GaussVolPointBase3D.C L759-768 / L856-865: such faces take dfdn = nf * snGrad [L945-948, L976-979]"""
    g = rc.load("gvp_other")
    assert set(g["nv"]) == {5, 6}
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        st, gs = om.fvsc("GaussVolPoint", "grad_s", g["cell_s"][i], np.zeros(0))
        st2, gv = om.fvsc("GaussVolPoint", "grad_v", g["cell_v"][i], np.zeros(0))
        assert st == 0 and st2 == 0
        assert rel(gs[0], g["grad_s"][i]) <= TOL and rel(gv[0], g["grad_v"][i]) <= TOL, (i, nv)
        om.close()


def qgdlength_meshes():
    g = rc.load("qgdlength")
    for mi in range(int(g["nMeshes"])):
        prim = {k: g[f"m{mi}_{k}"] for k in ("points", "faceOffsets", "facePoints", "owner", "neighbour", "patchStart", "patchSize", "patchType")}
        prim["nCells"] = int(g[f"m{mi}_nCells"])
        yield mi, prim, g[f"m{mi}_hQGDf"], g[f"m{mi}_hQGD"], g[f"m{mi}_hQGDb"]


def test_qgd_length_scales_of_whole_meshes():
    """QGDCoeffs::updateQGDLength [QGDCoeffs.C L298-376] executed from the listing text over three whole meshes (hexahedra, split
    quads, a one-cell-thick plane with empty patches): hQGDf, the area-weighted hQGD of every cell, hQGD on the patches"""
    for mi, prim, hf, hc, hb in qgdlength_meshes():
        om = OracleMesh(prim)
        oc = OracleCase(om, q.default_options(stencil="reduced"))
        n = prim["nCells"]
        for ip, t in enumerate(prim["patchType"]):
            if t == 1:
                oc.set_bc(ip, U=("none", None), T=("none", None), p=("none", None))
        oc.set_fields(np.zeros((n, 3)), np.ones(n), np.ones(n))
        assert rel(oc.field("hQGDf"), hf) <= TOL, mi
        assert rel(oc.field("hQGD"), hc) <= TOL, mi
        assert rel(oc.field("hQGD.boundary"), hb) <= TOL, mi
        oc.close(); om.close()


def courant_case(g, i, adjust=1):
    opt = case_options(g, i)
    opt.adjustTimeStep = adjust
    opt.maxCo, opt.maxDeltaT, opt.cTau = float(g["maxCo"][i]), float(g["maxDeltaT"][i]), float(g["cTau"][i])
    return opt


def test_courant_number_time_step_and_speed_of_sound():
    """QGDCourantNo.H L36-53, setDeltaT-QGDQHD.H L41-61 and c = sqrt(gamma/psi) [hePsiQGDThermo.C L123-124] from the listing text"""
    g = rc.load("courant")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        oc = OracleCase(om, courant_case(g, i))
        oc.set_fields(g["U"][i], g["T"][i], g["p"][i])
        assert rel(oc.field("c"), g["c_cells"][i]) <= TOL, i
        oc.step(1)
        info = oc.info()
        assert rel(info["CoNum"], g["CoNum"][i]) <= 1e-12 and rel(info["deltaT"], g["deltaT1"][i]) <= 1e-12, (i, info, g["CoNum"][i], g["deltaT1"][i])
        oc.close(); om.close()


def qhd_closure_options(g, i):
    from qgdsolver_amd import qhdfoam
    return qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel=int(g["model"][i]), Tau=float(g["Tau"][i]), aQGD=float(g["aQGD"][i]),
                               UQHD=float(g["UQHD"][i]), T0=float(g["T0"][i]), Gr=float(g["Gr"][i]), mu=float(g["mu"][i]), rho0=float(g["rho0"][i]))


def test_qhd_tau_closures():
    """constTau.C L73-74, HbyUQHD.C L82-83, T0byGr.C L86-87, H2bynuQHD.C L80-82 from the listing text"""
    from oracle import OracleQhdCase
    g = rc.load("qhdclosure")
    assert set(g["model"]) == {0, 1, 2, 3}
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        oc = OracleQhdCase(om, qhd_closure_options(g, i))
        oc.set_fields(np.zeros((2, 3)), np.full(2, 300.0), np.zeros(2))
        assert rel(oc.field("tauQGDf")[0], g["tauQGDf"][i]) <= TOL, (i, int(g["model"][i]), oc.field("tauQGDf")[0], g["tauQGDf"][i])
        oc.close(); om.close()


BND_FACE_FIELDS = ("tauQGDf", "gradUf", "gradef", "gradRhof", "phiwStar", "gradPf", "phiJm", "phi", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU")


def boundary_case(g, i, make_mesh, make_case):
    """the one-boundary-face configuration of ref_expr_casebnd: patch 0 with U fixedValue, T zeroGradient, p qgdFlux"""
    nv = int(g["nv"][i])
    mesh = make_mesh(*rc.boundary_face_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
    case = make_case(mesh, case_options_of(g, i))
    case.set_bc(0, U=("fixedValue", tuple(g["Ub"][i])), T=("zeroGradient", None), p=("qgdFlux", None))
    far = np.array([[0.0, 0.0, 0.0]])
    case.set_fields(np.vstack([g["U"][i][None, :], far + g["U"][i]]), np.array([g["T"][i], g["T"][i]]), np.array([g["p"][i], g["p"][i]]))
    return mesh, case


def case_options_of(g, i):
    return q.default_options(stencil="GaussVolPoint", R=float(g["R"][i]), Cv=float(g["Cv"][i]), mu=float(g["mu"][i]), Pr=float(g["Pr"][i]),
                             ScQGD=float(g["ScQGD"][i]), PrQGD=float(g["PrQGD"][i]), alphaQGD=float(g["alphaQGD"][i]), deltaT=1e-3)


def test_flux_assembly_of_one_boundary_face_with_the_qgdflux_condition():
    """updateFields.H / updateFluxes.H on a patch face, the boundary-face text of GaussVolPointBase3D.C for the four gradients,
    qgdFluxFvPatchScalarField::updateCoeffs [L184-192] evaluated inside fvsc::grad(p) with the fresh phiwStar (reference quirk B6),
    constScPrModel1's patch loop: all from the listing text (tests/golden/ref_expr_casebnd.npz)"""
    g = rc.load("casebnd")
    for i in range(len(g["nv"])):
        om, oc = boundary_case(g, i, oracle_mesh, OracleCase)
        oc.updateFluxes()
        for f in BND_FACE_FIELDS:
            assert rel(oc.field(f)[1], g[f][i]) <= 5e-13, (i, f, oc.field(f)[1], g[f][i])
        assert rel(oc.field("p.boundary")[0], g["pMid"][i]) <= TOL, (i, oc.field("p.boundary")[0], g["pMid"][i])
        oc.close(); om.close()


def qhd_eqn_options(g, i):
    from qgdsolver_amd import qhdfoam
    return qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="constTau", Tau=float(g["Tau"][i]), rho0=float(g["rho0"][i]), mu=float(g["mu"][i]),
                               Pr=float(g["Pr"][i]), beta=float(g["beta"][i]), g=tuple(g["g"][i]), deltaT=float(g["deltaT"][i]), pTol=1e-14,
                               pMaxIter=200, pRefCell=int(g["pRefCell"][i]), pRefValue=float(g["pRefValue"][i]), precond=0)


def test_one_whole_qhdfoam_step_from_the_listing_text():
    """QHDpEqn.H L35-47 (the pressure equation, setReference, flux()), QHDUEqn.H L36-84 and QHDTEqn.H L65-91 (explicit branch), the
    reference level of QHDFoam.C L123-130, on top of updateFields.H / updateFluxes.H: one step of the QHD case on a two-cell mesh"""
    from oracle import OracleQhdCase
    g = rc.load("qhdeqn")
    assert set(g["pRefCell"]) == {0, 1}
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        assert rel(om.array("nonOrthDeltaCoeffs")[0], g["delta"][i]) <= 1e-14
        oc = OracleQhdCase(om, qhd_eqn_options(g, i))
        oc.set_fields(g["U"][i], g["T"][i], g["p"][i])
        oc.step(1)
        for f, want in (("phiu", "phiu"), ("phiwo", "phiwo")):
            assert rel(oc.field(f)[0], g[want][i]) <= 1e-11, (i, f, oc.field(f)[0], g[want][i])
        # two cells closed by one face: the only divergence-free face flux is zero, so phi = phiu - phiwo + pEqn.flux() pins the SIGN of
        # the pressure correction (p_N - p_O = (phiu - phiwo)/a), not a value
        assert abs(oc.field("phi")[0] - g["phi1"][i]) <= 1e-11 * max(abs(g["phiu"][i]), abs(g["phiwo"][i])), (i, oc.field("phi")[0], g["phi1"][i])
        for f, want in (("p", "p1"), ("U", "U1"), ("T", "T1")):
            assert rel(oc.field(f), g[want][i]) <= 1e-11, (i, f, oc.field(f), g[want][i])
        oc.close(); om.close()


def test_one_whole_qhdfoam_step_implicit_branch_from_the_listing_text():
    """the same step through the listing's own `if (implicitDiffusion)` [QHDUEqn.H L46-65, QHDTEqn.H L69-80]: fvm::laplacian(muf/rhof, U)
    and fvm::laplacian(Hif, T) as matrices on the two-cell mesh (tests/golden/ref_expr_qhdeqn_implicit.npz)"""
    from oracle import OracleQhdCase
    g = rc.load("qhdeqn_implicit")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        opt = qhd_eqn_options(g, i)
        opt.implicitDiffusion, opt.implicitTol, opt.implicitMaxIter = 1, 1e-15, 100
        oc = OracleQhdCase(om, opt)
        oc.set_fields(g["U"][i], g["T"][i], g["p"][i])
        oc.step(1)
        assert abs(oc.field("phi")[0] - g["phi1"][i]) <= 1e-11 * max(abs(g["phiu"][i]), abs(g["phiwo"][i])), i
        for f, want in (("p", "p1"), ("U", "U1"), ("T", "T1")):
            assert rel(oc.field(f), g[want][i]) <= 1e-11, (i, f, oc.field(f), g[want][i])
        # and the branch matters on these inputs: the explicit oracle lands elsewhere
        opt.implicitDiffusion = 0
        ex = OracleQhdCase(om, opt)
        ex.set_fields(g["U"][i], g["T"][i], g["p"][i])
        ex.step(1)
        assert rel(ex.field("U"), g["U1"][i]) > 1e-9, i
        oc.close(); ex.close(); om.close()


def test_one_step_of_the_implicit_diffusion_branch_from_the_listing_text():
    """QGDUEqn.H L36-75 and QGDEEqn.H L37-64 with implicitDiffusion true, executed from the listing text on the two-cell mesh of
    case2cell: the explicit part with the fluxes of the implicit branch, the implicit U and e solves (fvm::ddt(rho, .) - fvc::ddt(rho, .)
    - fvm::laplacian), rhoU = rho U, phiSigmaDotU = Sf & ((muf lin(grad U) + tauMC) & Uf) from the new velocity, rhoE = rho (e + |U|^2/2)"""
    g = rc.load("implicit2cell")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        om = oracle_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        assert rel(om.array("nonOrthDeltaCoeffs")[0], g["delta"][i]) <= 1e-14
        opt = case_options(g, i)
        opt.implicitDiffusion, opt.implicitTol, opt.implicitMaxIter = 1, 1e-15, 100
        oc = OracleCase(om, opt)
        oc.set_fields(g["U"][i], g["T"][i], g["p"][i])
        oc.step(1)
        for f in ("rho", "U", "e", "rhoE"):
            assert rel(oc.field(f), g[f + "1"][i]) <= 1e-11, (i, nv, f, oc.field(f), g[f + "1"][i])
        assert rel(oc.field("phiSigmaDotU")[0], g["phiSigmaDotU"][i]) <= 1e-10, (i, oc.field("phiSigmaDotU")[0], g["phiSigmaDotU"][i])
        oc.close(); om.close()


def test_leastsquares_stencil_order_from_the_listing_text():
    """extendedFaceStencilFindNeighbours.C L48-84 executed as listed on three 2-D meshes: which cells every internal face gathers and
    in which order (the order of the weights' and the gradient's sums)"""
    from util import make_mesh, oracle_mesh_of
    g = rc.load("lsqorder")
    for kind in ("plane2d_jitter", "step2d", "plane2d"):
        mesh = make_mesh(kind)
        om = oracle_mesh_of(mesh)
        off, cells = g[kind + "_off"], g[kind + "_cells"]
        assert len(off) == mesh.nInternalFaces + 1
        for f in range(mesh.nInternalFaces):
            assert om.lsq_stencil(f) == [int(c) for c in cells[off[f]:off[f + 1]]], (kind, f)


def qhdflux_case(case_cls, mesh_handle, mesh):
    """the small buoyant cavity of ref_expr_qhdflux (walls with the qhdFlux pressure condition fed by the registered flux)"""
    from test_qhd_case import cavity_bcs, initial, options
    c = case_cls(mesh_handle, options(deltaT=1e-3))
    cavity_bcs(c, mesh)
    U, T, p = initial(mesh)
    U = U + 1e-2 * np.random.default_rng(3).standard_normal(U.shape)
    c.set_fields(U, T, p)
    return c


def test_qhdflux_wall_gradient_from_the_listing_text():
    """qhdFluxFvPatchScalarField.C L193-203 (updateCoeffs with the registered flux) + fixedGradient's evaluate: the patch pressure of
    the walls after three steps is that formula of the step's own phiwo"""
    from oracle import OracleQhdCase
    from util import make_mesh, oracle_mesh_of
    g = rc.load("qhdflux")
    mesh = make_mesh("box654_jitter")
    oc = qhdflux_case(OracleQhdCase, oracle_mesh_of(mesh), mesh)
    oc.step(int(g["steps"]))
    nif = mesh.nInternalFaces
    assert np.abs(g["gradient"]).max() > 1e-6           # the walls do carry a gradient
    assert rel(oc.field("phiwo")[nif:], g["phiwo_b"]) <= 1e-13
    assert rel(oc.field("p.boundary"), g["pb"]) <= 1e-12


def species_equation_inputs(g, i):
    """arguments of qgdfoam.QGDYEqn for case i of ref_expr_specieseqn (the two-cell mesh has no boundary faces)"""
    ns = g["Y"].shape[1]
    Y = [(g["Y"][i][k].copy(), np.zeros(0)) for k in range(ns)]
    jm = [np.array([g["phiJmY"][i][k]]) for k in range(ns)]
    df = [np.array([g["diffusiveFlux0"][i][k]]) for k in range(ns)]
    Su = [g["Su"][i][k].copy() for k in range(ns)]
    return Y, jm, df, Su, np.array([float(g["muf"][i])])


def test_species_equation_from_the_listing_text():
    """QGDYEqn.H L40-45, L69-92 executed as listed (three species, the last inert, explicit sources, one case with a value the step
    drives below zero) against qgdfoam.QGDYEqn over the oracle's orc_species_step"""
    from qgdsolver_amd import qgdfoam
    from test_qhd_pressure import HostDev
    g = rc.load("specieseqn")
    assert (g["Ynew"] == 0.0).any()
    for i in range(len(g["nv"])):
        prim, geom = rc.two_cell_mesh(g["pts"][i], int(g["nv"][i]), g["Sf"][i], g["Cf"][i], g["C"][i])
        om = oracle_mesh(prim, geom)
        assert rel(om.array("nonOrthDeltaCoeffs")[0], g["delta"][i]) <= 1e-14
        mesh = type("M", (), dict(nCells=2, nFaces=1, nBoundaryFaces=0))()
        Y, jm, df, Su, muf = species_equation_inputs(g, i)

        def call(*a):
            assert oracle.species_step(om, *a) == 0
        new = qgdfoam.QGDYEqn(HostDev(mesh), Y, g["rhoOld"][i], g["rho"][i], jm, muf, list(g["Sc"][i]), float(g["deltaT"][i]), df,
                              int(g["inertIndex"][i]), Su=Su, call=call)
        for k in range(len(new)):
            assert np.abs(new[k] - g["Ynew"][i][k]).max() <= 1e-12, (i, k, new[k], g["Ynew"][i][k])
            assert abs(df[k][0] - g["diffusiveFlux1"][i][k]) <= 1e-12 * max(1.0, abs(g["diffusiveFlux1"][i][k])), (i, k)
        om.close()
