"""The HIP path against numbers evaluated mechanically from the reference's own listing text (tests/golden/make_ref_expr.py),
through the C-ABI: qgd_mesh_create + qgd_mesh_set_geometry, the fvsc operators, and the QGDFoam case (whose internal face
goes through the fused, loads-first face kernel).  Same golden files and meshes as tests/test_ref_expr.py uses for the oracle."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import fvsc

import ref_expr_cases as rc
from test_ref_expr import FACE_FIELDS, QHD_FIELDS, STEP_FIELDS, bnd_ops, lsq_bnd_mesh, case_options, qhd_inputs, rel

pytestmark = pytest.mark.gpu

TOL = 1e-12


def device_mesh(prim, geom):
    m = q.PolyMesh.from_arrays(prim["points"], prim["faceOffsets"], prim["facePoints"], prim["owner"], prim["neighbour"], prim["nCells"],
                               prim["patchStart"], prim["patchSize"], prim["patchType"])
    m.set_geometry(geom["Sf"], geom["Cf"], geom["C"], geom["V"])
    return m


def test_gaussvolpoint_3d_on_the_device():
    g = rc.load("gvp3d")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}})
        gs = fvsc.grad(dev, q.volField("f", g["cell_s"][i], np.zeros(0)))
        gv = fvsc.grad(dev, q.volField("U", g["cell_v"][i], np.zeros((0, 3))))
        assert rel(gs[0], g["grad_s"][i]) <= TOL, (i, nv)
        assert rel(gv[0], g["grad_v"][i]) <= TOL, (i, nv)
        dv = fvsc.div(dev, q.volField("U", g["cell_v"][i], np.zeros((0, 3))))
        dt = fvsc.div(dev, q.volField("T", g["cell_t"][i], np.zeros((0, 9))))
        assert rel(dv[0], g["div_v"][i]) <= TOL, (i, nv)
        assert rel(dt[0], g["div_t"][i]) <= TOL, (i, nv)
        dev.close()


def test_gaussvolpoint_3d_boundary_faces_on_the_device():
    """the boundary-face text of GaussVolPointBase3D.C (tests/golden/ref_expr_gvp3d_bnd.npz) through qgd_fvsc_*"""
    g = rc.load("gvp3d_bnd")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.boundary_face_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}})
        for op, cell, bnd, want in bnd_ops(g, i):
            vf = q.volField("f", np.array(cell, float), np.array(bnd, float))
            got = fvsc.grad(dev, vf) if op.startswith("grad") else fvsc.div(dev, vf)
            assert rel(got[1], want) <= TOL, (i, nv, op, got[1], want)
        dev.close()


def test_gaussvolpoint_2d_on_the_device():
    g = rc.load("gvp2d")
    for i in range(len(g["ie3"])):
        ie3 = int(g["ie3"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], empty_normals=[rc.unit(ie3)]))
        assert mesh.nGeometricD == 2
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}})
        gs = fvsc.grad(dev, q.volField("f", g["f"][i], np.zeros(1)))
        assert rel(gs[0], g["grad"][i]) <= TOL, (i, ie3)
        dev.close()


def test_gaussvolpoint_2d_boundary_faces_on_the_device():
    g = rc.load("gvp2d_bnd")
    for i in range(len(g["ie3"])):
        ie3 = int(g["ie3"][i])
        mesh = device_mesh(*rc.boundary_face_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], back_axis=(ie3 + 1) % 3,
                                                  empty_normals=[rc.unit(ie3)]))
        assert mesh.nGeometricD == 2
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}})
        gs = fvsc.grad(dev, q.volField("f", np.array([g["f"][i], 0.3]), np.array([g["fb"][i], 0.0])))
        assert rel(gs[1], g["grad"][i]) <= TOL, (i, ie3, gs[1], g["grad"][i])
        dev.close()


def test_reduced_stencil_on_the_device():
    g = rc.load("reduced")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "reduced"}})
        for op, cell, nb in (("grad_s", g["cell_s"][i], ()), ("grad_v", g["cell_v"][i], (3,)), ("div_v", g["cell_v"][i], (3,)), ("div_t", g["cell_t"][i], (9,))):
            vf = q.volField("f", cell, np.zeros((0,) + nb))
            got = fvsc.grad(dev, vf) if op.startswith("grad") else fvsc.div(dev, vf)
            assert rel(got[0], g[op][i]) <= TOL, (i, nv, op)
        dev.close()


def test_leastsquares_boundary_faces_on_the_device():
    g = rc.load("lsq_bnd")
    for i in range(len(g["ie3"])):
        mesh = device_mesh(*lsq_bnd_mesh(g, i))
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "leastSquares"}})
        gs = fvsc.grad(dev, q.volField("f", np.array([g["f"][i], 0.3]), np.array([g["fb"][i], 0.0])))
        if g["symmetry"][i]:
            assert np.array_equal(gs[1], np.zeros(3))
        else:
            assert rel(gs[1], g["grad"][i]) <= TOL, (i, gs[1], g["grad"][i])
        dev.close()


def test_leastsquares_on_the_device():
    g = rc.load("lsq")
    for i in range(len(g["n"])):
        n = int(g["n"][i])
        prim, geom = rc.lsq_mesh(n, g["Cf"][i], g["centres"][i], one_d=(n == 2))
        mesh = device_mesh(prim, geom)
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "leastSquares", "grad(r)": "reduced"}})
        nb = mesh.nBoundaryFaces
        got = fvsc.grad(dev, q.volField("f", g["iF"][i][:n], np.zeros(nb)))
        if g["deg"][i]:
            red = fvsc.grad(dev, q.volField("r", g["iF"][i][:n], np.zeros(nb)))
            assert rel(got[0], red[0]) <= 1e-15, i
        else:
            assert rel(got[0], g["grad"][i]) <= TOL, (i, n)
        dev.close()


def test_flux_assembly_of_one_face_on_the_device():
    g = rc.load("case2cell")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        case = q.QGDFoamCase(dev, case_options(g, i))
        case.set_fields(g["U"][i], g["T"][i], g["p"][i])
        case.updateFluxes()
        for f in FACE_FIELDS:
            if f in ("rhof", "Uf", "pf", "cf", "Hf", "alphauf", "muf"):
                continue   # interpolated fields stay in registers on the device; their consumers below are compared
            assert rel(case.field(f)[0], g[f][i]) <= TOL, (i, nv, f, case.field(f)[0], g[f][i])
        for f in ("muQGD", "alphauQGD", "tauQGD", "hQGD"):
            assert rel(case.field(f), g[f][i]) <= TOL, (i, f)
        # one explicit step against QGDRhoEqn.H / QGDUEqn.H / QGDEEqn.H executed from the listing text
        case.step(1)
        for f in STEP_FIELDS:
            assert rel(case.field(f), g[f + "1"][i]) <= TOL, (i, nv, f, case.field(f), g[f + "1"][i])
        th = rc.load("thermo2cell")     # thermo.correct(), p = rho / psi: hePsiQGDThermo.C L48-64, L123-124, QGDFoam.C L152-154 from the text
        for f in ("T", "p", "c"):
            assert rel(case.field(f), th[f + "1"][i]) <= TOL, (i, nv, f, case.field(f), th[f + "1"][i])
        case.close(); dev.close()


def test_implicit_branch_face_expressions_on_the_device():
    """updateFluxes.H with implicitDiffusion true against the listing text: phiPi, phiQ from the face kernel, phiTauMC as the implicit
    step forms it from fvc::grad(U) of the state before the step"""
    g = rc.load("case2cell")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        opt = case_options(g, i)
        opt.implicitDiffusion = 1
        case = q.QGDFoamCase(dev, opt)
        case.set_fields(g["U"][i], g["T"][i], g["p"][i])
        case.updateFluxes()
        for f, want in (("phiPi", "phiPi_impl"), ("phiQ", "phiQ_impl"), ("phiJm", "phiJm")):
            assert rel(case.field(f)[0], g[want][i]) <= TOL, (i, nv, f, case.field(f)[0], g[want][i])
        case.step(1)
        assert rel(case.field("phiTauMC")[0], g["phiTauMC"][i]) <= TOL, (i, nv, case.field("phiTauMC")[0], g["phiTauMC"][i])
        case.close(); dev.close()


def test_qhd_face_expressions_on_the_device():
    """qgd_qhd_fluxes against the QHDFoam face expressions evaluated from the listing text (tests/golden/ref_expr_qhdface.npz)"""
    from qgdsolver_amd import qhdfoam
    g = rc.load("qhdface")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        a = qhd_inputs(g, i)
        res = qhdfoam.updateFluxes(dev, "GaussVolPoint", a["U"], a["T"], a["rho"], a["tauQGDf"], a["beta"], a["g"], p=a["p"], phi=a["phi"])
        for f in QHD_FIELDS:
            assert rel(res[f][0], g[f][i]) <= TOL, (i, nv, f, res[f][0], g[f][i])
        dev.close()


def test_species_flux_expressions_on_the_device():
    """qgd_species_flux against reactingLagrangianQGDFoam/updateFluxes.H L122-127 evaluated from the listing text"""
    from qgdsolver_amd import qgdfoam
    g = rc.load("species")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        res = qgdfoam.speciesFlux(dev, "GaussVolPoint", (g["Y"][i], np.zeros(0)), (g["U"][i], np.zeros((0, 3))), [g["phiJm"][i]], [g["phi"][i]],
                                  [g["tauQGDf"][i]])
        for f in ("gradYf", "phiJmY", "diffusiveFlux"):
            assert rel(res[f][0], g[f][i]) <= TOL, (i, nv, f, res[f][0], g[f][i])
        dev.close()


# ---- round 3: the sections that were still outside the mechanical pin ------------------------------------------------------
def test_gaussvolpoint_2d_vector_operators_on_the_device():
    g = rc.load("gvp2d_vec")
    for i in range(len(g["ie3"])):
        ie3 = int(g["ie3"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], 4, g["Sf"][i], g["Cf"][i], g["C"][i], empty_normals=[rc.unit(ie3)]))
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}})
        gv = fvsc.grad(dev, q.volField("U", g["U"][i], np.zeros((1, 3))))
        dv = fvsc.div(dev, q.volField("U", g["U"][i], np.zeros((1, 3))))
        dt = fvsc.div(dev, q.volField("T", g["Tn"][i], np.zeros((1, 9))))
        assert rel(gv[0], g["grad_v"][i]) <= TOL and rel(dv[0], g["div_v"][i]) <= TOL and rel(dt[0], g["div_t"][i]) <= TOL, (i, ie3)
        dev.close()


def test_faces_with_more_than_four_vertices_on_the_device():
    g = rc.load("gvp_other")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}})
        gs = fvsc.grad(dev, q.volField("f", g["cell_s"][i], np.zeros(0)))
        gv = fvsc.grad(dev, q.volField("U", g["cell_v"][i], np.zeros((0, 3))))
        assert rel(gs[0], g["grad_s"][i]) <= TOL and rel(gv[0], g["grad_v"][i]) <= TOL, (i, nv)
        dev.close()


def test_qgd_length_scales_on_the_device():
    from test_ref_expr import qgdlength_meshes
    for mi, prim, hf, hc, hb in qgdlength_meshes():
        mesh = q.PolyMesh.from_arrays(prim["points"], prim["faceOffsets"], prim["facePoints"], prim["owner"], prim["neighbour"], prim["nCells"],
                                      prim["patchStart"], prim["patchSize"], prim["patchType"])
        dev = q.Device(mesh)
        assert rel(fvsc.device_field(dev, "hQGDf"), hf) <= TOL, mi
        assert rel(fvsc.device_field(dev, "hQGD"), hc) <= TOL, mi
        assert rel(fvsc.device_field(dev, "hQGD.boundary"), hb) <= TOL, mi
        dev.close()


def test_courant_number_and_time_step_on_the_device():
    from test_ref_expr import courant_case
    g = rc.load("courant")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        case = q.QGDFoamCase(dev, courant_case(g, i))
        case.set_fields(g["U"][i], g["T"][i], g["p"][i])
        assert rel(case.field("c"), g["c_cells"][i]) <= TOL, i
        case.step(1)
        info = case.info()
        assert rel(info["CoNum"], g["CoNum"][i]) <= 1e-12 and rel(info["deltaT"], g["deltaT1"][i]) <= 1e-12, (i, info)
        case.close(); dev.close()


def test_qhd_tau_closures_on_the_device():
    from qgdsolver_amd import qhdfoam
    from test_ref_expr import qhd_closure_options
    g = rc.load("qhdclosure")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        case = qhdfoam.QHDFoamCase(dev, qhd_closure_options(g, i))
        case.set_fields(np.zeros((2, 3)), np.full(2, 300.0), np.zeros(2))
        assert rel(case.field("tauQGDf")[0], g["tauQGDf"][i]) <= TOL, (i, int(g["model"][i]))
        case.close(); dev.close()


def test_flux_assembly_of_one_boundary_face_on_the_device():
    """the boundary-face kernels (boundaryFaceFluxKernel incl. the mid-step evaluation of the qgdFlux condition, quirk B6) against
    tests/golden/ref_expr_casebnd.npz: the listing text of updateFields.H, updateFluxes.H, GaussVolPointBase3D.C's boundary faces,
    qgdFluxFvPatchScalarField.C L184-192 and constScPrModel1.C's patch loop"""
    from test_ref_expr import BND_FACE_FIELDS, boundary_case
    g = rc.load("casebnd")
    for i in range(len(g["nv"])):
        mesh, case = boundary_case(g, i, device_mesh, lambda m, opt: q.QGDFoamCase(q.Device(m), opt))
        case.updateFluxes()
        for f in BND_FACE_FIELDS:
            if f in ("gradef", "gradRhof"):   # T zeroGradient, p_b = p_O at start-up: e_b = e_O, rho_b = rho_O, the gradients are rounding noise around zero
                assert np.abs(case.field(f)[1] - g[f][i]).max() <= 1e-12, (i, f)
                continue
            if f == "phiQ":   # with those two gradients gone the heat flux is noise too: held against the energy flux it travels with
                assert abs(case.field(f)[1] - g[f][i]) <= 2e-12 * abs(g["phiJmH"][i]), (i, f)
                continue
            assert rel(case.field(f)[1], g[f][i]) <= 2e-12, (i, f, case.field(f)[1], g[f][i])
        assert rel(case.field("p.boundary")[0], g["pMid"][i]) <= TOL, i
        dev = case.dev
        case.close(); dev.close()


def test_one_whole_qhdfoam_step_on_the_device():
    """the resident QHD case (qgd_qhd_case_step: flux assembly, pressure equation, U and T equations, reference level) against
    tests/golden/ref_expr_qhdeqn.npz: QHDpEqn.H, QHDUEqn.H, QHDTEqn.H and QHDFoam.C L123-130 executed from the listing text"""
    from qgdsolver_amd import qhdfoam
    from test_ref_expr import qhd_eqn_options
    g = rc.load("qhdeqn")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        case = qhdfoam.QHDFoamCase(dev, qhd_eqn_options(g, i))
        case.set_fields(g["U"][i], g["T"][i], g["p"][i])
        case.step(1)
        for f, want in (("phiu", "phiu"), ("phiwo", "phiwo")):
            assert rel(case.field(f)[0], g[want][i]) <= 1e-11, (i, f)
        assert abs(case.field("phi")[0] - g["phi1"][i]) <= 1e-11 * max(abs(g["phiu"][i]), abs(g["phiwo"][i])), i
        for f, want in (("p", "p1"), ("U", "U1"), ("T", "T1")):
            assert rel(case.field(f), g[want][i]) <= 1e-10, (i, f, case.field(f), g[want][i])
        case.close(); dev.close()


def test_one_whole_qhdfoam_step_implicit_branch_on_the_device():
    """the resident QHD case with implicitDiffusion true (the four systems {Ux, Uy, Uz, T} as one multi-right-hand-side solve) against
    tests/golden/ref_expr_qhdeqn_implicit.npz: QHDUEqn.H L46-65 and QHDTEqn.H L69-80 executed from the listing text"""
    from qgdsolver_amd import qhdfoam
    from test_ref_expr import qhd_eqn_options
    g = rc.load("qhdeqn_implicit")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        opt = qhd_eqn_options(g, i)
        opt.implicitDiffusion, opt.implicitTol, opt.implicitMaxIter = 1, 1e-15, 200
        case = qhdfoam.QHDFoamCase(dev, opt)
        case.set_fields(g["U"][i], g["T"][i], g["p"][i])
        case.step(1)
        assert abs(case.field("phi")[0] - g["phi1"][i]) <= 1e-11 * max(abs(g["phiu"][i]), abs(g["phiwo"][i])), i
        for f, want in (("p", "p1"), ("U", "U1"), ("T", "T1")):
            assert rel(case.field(f), g[want][i]) <= 1e-10, (i, f, case.field(f), g[want][i])
        case.close(); dev.close()


def test_one_step_of_the_implicit_diffusion_branch_on_the_device():
    """the device's implicitDiffusion step (qgd_implicit.hip: face terms, the four PCG solves, phiSigmaDotU) against
    tests/golden/ref_expr_implicit2cell.npz: QGDUEqn.H L36-75 and QGDEEqn.H L37-64 executed from the listing text"""
    g = rc.load("implicit2cell")
    for i in range(len(g["nv"])):
        nv = int(g["nv"][i])
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], nv, g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        opt = case_options(g, i)
        opt.implicitDiffusion, opt.implicitTol, opt.implicitMaxIter = 1, 1e-15, 100
        case = q.QGDFoamCase(dev, opt)
        case.set_fields(g["U"][i], g["T"][i], g["p"][i])
        case.step(1)
        for f in ("rho", "U", "e", "rhoE"):
            assert rel(case.field(f), g[f + "1"][i]) <= 1e-10, (i, nv, f, case.field(f), g[f + "1"][i])
        assert rel(case.field("phiSigmaDotU")[0], g["phiSigmaDotU"][i]) <= 1e-9, (i, case.field("phiSigmaDotU")[0], g["phiSigmaDotU"][i])
        case.close(); dev.close()


def test_leastsquares_stencil_order_on_the_device():
    """the sliced-ELL stencil lists the device sums over hold the cells of every internal face in the order
    extendedFaceStencilFindNeighbours.C L48-84 produces (ref_expr_lsqorder: that text executed)"""
    from util import make_mesh
    g = rc.load("lsqorder")
    for kind in ("plane2d_jitter", "step2d"):
        mesh = make_mesh(kind)
        dev = q.Device(mesh)
        off, cells = g[kind + "_off"], g[kind + "_cells"]
        faces = range(mesh.nInternalFaces) if mesh.nInternalFaces < 150 else range(0, mesh.nInternalFaces, 7)
        for f in faces:
            assert dev.lsq_stencil(f) == [int(c) for c in cells[off[f]:off[f + 1]]], (kind, f)
        dev.close()


def test_qhdflux_wall_gradient_on_the_device():
    """the device's QHD case: patch pressure of the qhdFlux walls against qhdFluxFvPatchScalarField.C L193-203 executed from the text"""
    from qgdsolver_amd import qhdfoam
    from test_ref_expr import qhdflux_case
    from util import make_mesh
    g = rc.load("qhdflux")
    mesh = make_mesh("box654_jitter")
    dev = q.Device(mesh)
    gc = qhdflux_case(qhdfoam.QHDFoamCase, dev, mesh)
    gc.step(int(g["steps"]))
    nif = mesh.nInternalFaces
    assert rel(gc.field("phiwo")[nif:], g["phiwo_b"]) <= 1e-8
    assert rel(gc.field("p.boundary"), g["pb"]) <= 1e-8
    gc.close(); dev.close()


def test_species_equation_on_the_device():
    """qgd_species_step (through qgdfoam.QGDYEqn) against QGDYEqn.H L40-45, L69-92 executed from the listing text"""
    from qgdsolver_amd import qgdfoam
    from test_ref_expr import species_equation_inputs
    g = rc.load("specieseqn")
    for i in range(len(g["nv"])):
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], int(g["nv"][i]), g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        Y, jm, df, Su, muf = species_equation_inputs(g, i)
        new = qgdfoam.QGDYEqn(dev, Y, g["rhoOld"][i], g["rho"][i], jm, muf, list(g["Sc"][i]), float(g["deltaT"][i]), df, int(g["inertIndex"][i]), Su=Su)
        for k in range(len(new)):
            assert np.abs(new[k] - g["Ynew"][i][k]).max() <= 1e-12, (i, k, new[k], g["Ynew"][i][k])
            assert abs(df[k][0] - g["diffusiveFlux1"][i][k]) <= 1e-12 * max(1.0, abs(g["diffusiveFlux1"][i][k])), (i, k)
        dev.close()


def test_species_equation_implicit_branch_on_the_device():
    """qgd_species_step_implicit (through qgdfoam.QGDYEqn(implicitDiffusion=True)) against QGDYEqn.H L40-66, L86-92 executed from the
    listing text (tests/golden/ref_expr_specieseqn_implicit.npz): the solve, diffusiveFlux[i] += YEqn.flux(), the inert species"""
    from qgdsolver_amd import qgdfoam
    from test_ref_expr import species_equation_inputs
    g = rc.load("specieseqn_implicit")
    for i in range(len(g["nv"])):
        mesh = device_mesh(*rc.two_cell_mesh(g["pts"][i], int(g["nv"][i]), g["Sf"][i], g["Cf"][i], g["C"][i]))
        dev = q.Device(mesh)
        Y, jm, df, Su, muf = species_equation_inputs(g, i)
        new = qgdfoam.QGDYEqn(dev, Y, g["rhoOld"][i], g["rho"][i], jm, muf, list(g["Sc"][i]), float(g["deltaT"][i]), df, int(g["inertIndex"][i]), Su=Su,
                              implicitDiffusion=True, tolerance=1e-15, maxIter=100)
        for k in range(len(new)):
            assert np.abs(new[k] - g["Ynew"][i][k]).max() <= 1e-11, (i, k, new[k], g["Ynew"][i][k])
            assert abs(df[k][0] - g["diffusiveFlux1"][i][k]) <= 1e-11 * max(1.0, abs(g["diffusiveFlux1"][i][k])), (i, k)
        dev.close()
