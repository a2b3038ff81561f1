"""The fused step of the explicit branch (fusedFaceCellKernel, QGD_FUSED): a workgroup stages a block of <= 128 cells and the cells around it in
LDS, forms the block's vertex values, computes every internal face of its cells into LDS and advances those cells from there; neither vertex
values nor net fluxes of internal faces reach device memory.  Same arithmetic per vertex, face and cell, same summation orders as
pointInterpRecKernel + faceFluxGvp3(Tile)Kernel + cellUpdateKernel: the states must agree BIT FOR BIT with QGD_FUSED=0 -- on hexahedra, on
jittered meshes with triangles and polygon faces, scrambled numberings, with patches of every kind the step knows (their fluxes come from the
boundary kernel through device memory), the qgdFlux walls of the forward step's 3-D cousin, and over a long run.  Cases the fused kernel
does not serve (Courant-number control, implicit diffusion, per-term stencils) must say so and run the three kernels."""
import os

import numpy as np
import pytest

import qgdsolver_amd as q

import cases
from test_config5_gpu import c5_mesh

pytestmark = pytest.mark.gpu


def run(mesh, steps, fused, bc_fn=None, chunks=(None,), env=None, **opt):
    env = dict(env or {})
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        # "any": fused whatever the blocks look like (the default leaves meshes whose blocks come out small to the three kernels)
        dev = q.Device(mesh, fused_tables="any" if fused else False)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", **opt))
    if bc_fn:
        bc_fn(case)
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    info = case.fused_info()
    for n in chunks:
        case.step(steps if n is None else n)
    out = {n: case.field(n).copy() for n in ("rho", "U", "p", "e", "rhoE", "p.boundary", "U.boundary")}
    i = case.info()
    out["mins"] = np.array([i["minRho"], i["minE"], i["steps"], i["time"]])
    case.close(); dev.close()
    return out, info


def meshes():
    yield "hex 20^3", q.PolyMesh.box(20, 20, 20)
    yield "hex 37x11x5", q.PolyMesh.box(37, 11, 5)
    yield "hex 150x6x6", q.PolyMesh.box(150, 6, 6)
    yield "hex 8x4x4 (one block)", q.PolyMesh.box(8, 4, 4)
    yield "hex 3x2x2 (less than a wavefront of cells)", q.PolyMesh.box(3, 2, 2)
    yield "jitter + triangles + polygons, Morton order", c5_mesh(16, 8 ** 3, poly=True)
    tri = q.PolyMesh.box(9, 7, 5)
    tri.jitter(0.15, seed=3); tri.split_quads(3)
    yield "every third quad split, natural order", tri
    scr = q.PolyMesh.box(12, 10, 8)
    scr.renumber(np.random.default_rng(5).permutation(scr.nCells).astype(np.int32))
    yield "scrambled labels", scr


def equal(a, b, tag):
    for k in a:
        assert np.isfinite(a[k]).all(), (tag, k)
        assert np.array_equal(a[k], b[k]), (tag, k, np.abs(a[k] - b[k]).max())


def test_fused_step_is_bit_identical_to_the_two_kernels():
    for tag, mesh in meshes():
        h = 1.0 / 20
        for opt in (dict(deltaT=0.05 * h), dict(deltaT=0.05 * h, mu=2e-3, consistentEnergy=1)):
            a, ia = run(mesh, 7, False, **opt)
            b, ib = run(mesh, 7, True, **opt)
            assert not ia["fused"] and ib["fused"], (tag, ia, ib)
            assert ib["blocks"] >= (mesh.nCells + 127) // 128 and ib["ldsBytes"] <= 64 * 1024, ib
            assert ib["facesComputed"] >= mesh.nInternalFaces, ib
            equal(a, b, (tag, opt))


def test_fused_step_with_streamed_face_areas():
    """QGD_SGEO=0 (what a mesh with caller-supplied Sf gets): the kernel reads Sf instead of rebuilding it from the staged vertices"""
    for tag, mesh in list(meshes())[:1] + list(meshes())[5:7]:
        a, _ = run(mesh, 5, False, env={"QGD_SGEO": "0"}, deltaT=2e-3, mu=1e-3)
        b, ib = run(mesh, 5, True, env={"QGD_SGEO": "0"}, deltaT=2e-3, mu=1e-3)
        assert ib["fused"]
        equal(a, b, tag)
        c, _ = run(mesh, 5, True, deltaT=2e-3, mu=1e-3)      # and Sf from the vertices gives the same states as before
        d, _ = run(mesh, 5, False, deltaT=2e-3, mu=1e-3)
        equal(c, d, tag)


def test_fused_step_with_upwind_fluxes():
    """`div(phiJm,U|H) Gauss upwind`: the UPW instantiation of the fused kernel against the UPW face kernel + cell kernel"""
    for tag, mesh in list(meshes())[:1] + list(meshes())[5:6]:
        for opt in (dict(fluxSchemeU=1), dict(fluxSchemeH=1), dict(fluxSchemeU=1, fluxSchemeH=1)):
            a, ia = run(mesh, 6, False, deltaT=2e-3, mu=1e-3, **opt)
            b, ib = run(mesh, 6, True, deltaT=2e-3, mu=1e-3, **opt)
            assert ib["fused"] and not ia["fused"]
            equal(a, b, (tag, opt))
        lin, _ = run(mesh, 6, True, deltaT=2e-3, mu=1e-3)
        assert not np.array_equal(lin["U"], b["U"])        # (the scheme does something)


def test_fused_step_with_patches_of_every_kind_and_in_chunks():
    import test_case_parity_gpu as t
    mesh = q.PolyMesh.box(14, 9, 6)
    mesh.jitter(0.1, seed=11)
    for bc_fn in (t.mixed_box_bcs, None):
        a, _ = run(mesh, 0, False, bc_fn=bc_fn, chunks=(1, 2, 5), deltaT=5e-4, mu=1e-3)
        b, ib = run(mesh, 0, True, bc_fn=bc_fn, chunks=(1, 2, 5), deltaT=5e-4, mu=1e-3)
        assert ib["fused"]
        equal(a, b, bc_fn)


def test_fused_step_stays_bit_identical_over_a_long_run():
    mesh = q.PolyMesh.box(48, 48, 48)
    opt = dict(deltaT=0.1 / 48 / 1.3)
    a, _ = run(mesh, 300, False, **opt)
    b, ib = run(mesh, 300, True, **opt)
    assert ib["fused"] and ib["blocks"] == 48 ** 3 // 128      # 8x4x4 bricks
    assert ib["facesComputed"] == 864 * (3 * 128 + 80) - 2 * 3 * 48 * 48    # 464 per brick, less what the domain's six sides lack
    equal(a, b, "48^3 x 300")


def test_fused_step_under_courant_number_control():
    """adjustTimeStep: the new deltaT needs every face's Courant number before the first cell may advance, so the blocks stop at their cells'
    flux sums (fusedFaceCellKernel<..., ADJ>: max Cof / min tauQGDf per block, five sums per cell), faceReduce + deltaT follow, cellFinishKernel
    advances.  Against the three kernels under the same control: deltaT, time and Courant number agree to rounding step by step, the states
    too (<= 1e-13; the ADJ instantiation contracts the flux algebra's multiply-adds its own way), on hexahedra, a jittered mesh with
    triangles and polygons, walls of every kind"""
    import test_case_parity_gpu as t

    def run_adj(mesh, fused, bc_fn, steps):
        dev = q.Device(mesh, fused_tables="any" if fused else False)
        case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-4, adjustTimeStep=1, maxCo=0.3, maxDeltaT=1.0, mu=1e-3))
        assert case.fused_info()["fusedAdjust"] == fused and not case.fused_info()["fused"]
        if bc_fn:
            bc_fn(case)
        U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
        case.set_fields(U, T, p)
        hist = []
        for _ in range(steps):
            case.step(1)
            i = case.info()
            hist.append((i["deltaT"], i["time"], i["CoNum"]))
        out = {n: case.field(n).copy() for n in ("rho", "U", "p", "e", "rhoE", "p.boundary", "U.boundary")}
        i = case.info()
        out["mins"] = np.array([i["minRho"], i["minE"]])
        case.close(); dev.close()
        return out, np.array(hist)

    jit = q.PolyMesh.box(14, 9, 6).jitter(0.1, seed=11)
    for tag, mesh, bc_fn in (("hex 20^3", q.PolyMesh.box(20, 20, 20), None), ("jitter + walls", jit, t.mixed_box_bcs),
                             ("triangles + polygons", c5_mesh(16, 8 ** 3, poly=True), None)):
        a, ha = run_adj(mesh, False, bc_fn, 12)
        b, hb = run_adj(mesh, True, bc_fn, 12)
        assert np.abs(ha - hb).max() <= 1e-13 * np.abs(ha).max(), (tag, ha[-1], hb[-1])
        assert ha[-1, 0] > 1.5 * ha[0, 0]                      # (deltaT did grow under the control)
        for k in a:
            assert np.isfinite(b[k]).all() and np.abs(a[k] - b[k]).max() <= 1e-13 * np.abs(a[k]).max(), (tag, k, np.abs(a[k] - b[k]).max())


def test_cases_the_fused_kernel_does_not_serve_keep_the_two_kernels():
    mesh = q.PolyMesh.box(10, 8, 6)
    dev = q.Device(mesh, fused_tables="any")
    for opt in (dict(adjustTimeStep=1, maxCo=0.2), dict(implicitDiffusion=1, mu=1e-3), dict(termStencils={"grad(p)": "reduced"})):
        case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, **opt))
        fi = case.fused_info()
        assert not fi["fused"], opt
        assert fi["fusedAdjust"] == bool(opt.get("adjustTimeStep")), (opt, fi)   # (Courant-number control: blocks up to their sums + a cell kernel)
        # (the implicitDiffusion branch has its own block-fused assembly on the same blocks: tests/test_implicit_diffusion.py)
        assert fi["fusedImplicit"] == bool(opt.get("implicitDiffusion")), (opt, fi)
        case.close()
    case = q.QGDFoamCase(dev, q.default_options(stencil="reduced", deltaT=1e-3))
    assert not case.fused_info()["fused"]
    case.close()
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    assert case.fused_info()["fused"]
    # step_phase on such a case runs the two kernels in place; mixing the two ways of stepping is legal
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    case.step(2); case.step_phase(0); case.step_phase(1); case.step(1)
    ref = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, adjustTimeStep=0))
    ref.set_fields(U, T, p)
    for _ in range(4):
        ref.step_phase(0); ref.step_phase(1)
    for n in ("rho", "U", "p", "e"):
        assert np.array_equal(case.field(n), ref.field(n)), n
    case.close(); ref.close(); dev.close()


def shard_run(shard, fused, order, steps=4):
    dev = q.Device(shard, fused_tables="any" if fused else False)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3))
    U, T, p = cases.box_initial_fields(shard.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    assert case.fused_info()["fused"] == fused
    for _ in range(steps):          # no exchange: the ghost cells keep their values, in both runs
        for ph in order:
            case.step_phase(ph)
    case.sync()
    out = {n: case.field(n).copy() for n in ("rho", "U", "p", "e", "rhoE", "p.boundary", "U.boundary")}
    case.close(); dev.close()
    return out


def test_default_leaves_meshes_with_small_blocks_to_the_three_kernels():
    """QGD_FUSED unset (= 1): a box of bricks is fused; a mesh with every quadrilateral split (twelve faces per cell: a block holds 64 cells
    before its 512 faces are full) is not -- every block costs a workgroup two face passes whatever it holds"""
    assert "QGD_FUSED" not in os.environ
    tri = q.PolyMesh.box(20, 20, 20)
    tri.jitter(0.15, seed=3); tri.split_quads(1)
    for mesh, want in ((q.PolyMesh.box(32, 16, 16), True), (tri, False)):
        dev = q.Device(mesh)
        case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
        info = case.fused_info()
        assert info["fused"] == want, (mesh.nCells, info)
        case.close(); dev.close()


@pytest.mark.parametrize("nShards,which", [(2, 0), (2, 1), (3, 1)])
def test_fused_step_on_a_shard(nShards, which):
    """ghost cells belong to no block, the cells a neighbour waits for form the first blocks: phases 0, 1 and the boundary-layer-first order
    0, 10, 11 (the swap of the record buffers falls between 10 and 11) against the two kernels, bit for bit"""
    for mesh in (q.PolyMesh.box(24, 12, 12), c5_mesh(16, 8 ** 3, poly=True)):
        shard = mesh.shard(nShards, which)
        ref = shard_run(shard, False, (0, 1))
        for order in ((0, 1), (0, 10, 11)):
            got = shard_run(shard, True, order)
            equal(ref, got, (nShards, which, order))
