"""The LDS-staged face kernel (faceFluxGvp3TileKernel: distinct records of a 128-face tile loaded once as contiguous pieces,
picked out of LDS by every face) against the gather kernel it replaces (faceFluxGvp3Kernel, QGD_FTILE=0).  Same loads,
same arithmetic in the same order: the states must agree BIT FOR BIT -- on hexahedra, on jittered meshes with triangles and
polygon faces (the generic faces of a tile still go through the gather path inside the staged kernel), with Courant-number
control (the per-tile reductions), on a shard (ghost cells) and for every tile size.  Parity of either kernel with the oracle
is the business of test_case_parity_gpu.py / test_fullsize_gpu.py, which run the staged kernel by default.
"""
import os

import numpy as np
import pytest

import qgdsolver_amd as q

import cases
from test_config5_gpu import c5_mesh

pytestmark = pytest.mark.gpu


def run(mesh, steps, env, phases=False, **opt):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        dev = q.Device(mesh)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    ft = dev.face_tiles()
    if env.get("QGD_FTILE") == "0":
        assert ft["facesPerTile"] == 0
    elif ft["facesPerTile"]:   # the staged kernel runs, the gather kernel only mops up (at most a quarter of the tiles)
        assert ft["facesPerTile"] == int(env.get("QGD_FBLOCK", 128)) and 0 < ft["ldsBytes"] <= 65536, ft
        assert 4 * ft["gatherTiles"] <= ft["tiles"], ft
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", **opt))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    if phases:          # a shard: the two phases of a step, no exchange (the ghost cells keep their values)
        for _ in range(steps):
            case.step_phase(0)
            case.step_phase(1)
    else:
        case.step(steps)
    out = {n: case.field(n).copy() for n in ("rho", "U", "p", "e", "rhoE")}
    out["deltaT"] = np.array([case.info()["deltaT"]])
    out["tiles"] = ft
    case.close(); dev.close()
    return out


def meshes():
    yield "hex 20^3", q.PolyMesh.box(20, 20, 20)
    yield "hex 37x11x5 (ragged last tile)", q.PolyMesh.box(37, 11, 5)
    yield "hex 150x6x6 (long rows)", q.PolyMesh.box(150, 6, 6)
    yield "jitter + triangles + polygons, Morton order", c5_mesh(16, 8 ** 3, poly=True)
    tri = q.PolyMesh.box(9, 7, 5)
    tri.jitter(0.15, seed=3); tri.split_quads(3)
    yield "every third quad split, natural order", tri
    scr = q.PolyMesh.box(12, 10, 8)
    scr.renumber(np.random.default_rng(5).permutation(scr.nCells).astype(np.int32))
    yield "scrambled labels (most tiles beyond the caps)", scr
    rcm = q.PolyMesh.box(30, 6, 5)
    rcm.renumber(rcm.rcm_order())
    yield "reverse Cuthill-McKee order", rcm


@pytest.mark.parametrize("fb", [64, 128, 256])
def test_staged_kernel_is_bit_identical_to_gather_kernel(fb):
    for tag, mesh in meshes():
        h = 1.0 / 20
        for opt in (dict(deltaT=0.05 * h), dict(deltaT=0.05 * h, adjustTimeStep=1, maxCo=0.2)):
            a = run(mesh, 5, {"QGD_FTILE": "0", "QGD_FBLOCK": str(fb)}, **opt)
            b = run(mesh, 5, {"QGD_FTILE": "1", "QGD_FBLOCK": str(fb)}, **opt)
            ft = b.pop("tiles"); a.pop("tiles")
            if tag.startswith("hex 20") or tag.startswith("jitter"):
                assert ft["facesPerTile"] == fb, (tag, ft)    # these do go through the staged kernel
            if tag.startswith("hex 150") and fb <= 128:
                assert ft["gatherTiles"] > 0, ft     # the last row of a box: one internal face per cell, 4 fresh vertices each
            for k in a:
                assert np.isfinite(a[k]).all()
                assert np.array_equal(a[k], b[k]), (tag, fb, opt, k, np.abs(a[k] - b[k]).max())


def test_offset_table_and_fixed_stride_lists_agree():
    """The staged kernel reads its tile lists from the fixed-stride copy by default (offsets computed: one dependent round trip less per
    tile); QGD_FTILE_FIXED=0 keeps the offset table.  Both against the gather kernel, bit for bit, on every mesh of this module."""
    for tag, mesh in meshes():
        h = 1.0 / 20
        for opt in (dict(deltaT=0.05 * h), dict(deltaT=0.05 * h, adjustTimeStep=1, maxCo=0.2)):
            for sgeo in ("1", "0"):     # Sf of quadrilaterals rebuilt from the staged vertices / streamed (as with the caller's own geometry)
                a = run(mesh, 5, {"QGD_FTILE": "0", "QGD_SGEO": sgeo}, **opt)
                a.pop("tiles")
                for fixed in ("0", "1"):
                    b = run(mesh, 5, {"QGD_FTILE": "1", "QGD_FTILE_FIXED": fixed, "QGD_SGEO": sgeo}, **opt)
                    b.pop("tiles")
                    for k in a:
                        assert np.array_equal(a[k], b[k]), (tag, sgeo, fixed, opt, k, np.abs(a[k] - b[k]).max())


def test_staged_kernel_on_a_shard():
    mesh = q.PolyMesh.box(24, 12, 12)
    shard = mesh.shard(2, 1)
    a = run(shard, 3, {"QGD_FTILE": "0"}, phases=True, deltaT=1e-3, adjustTimeStep=1, maxCo=0.2)
    b = run(shard, 3, {"QGD_FTILE": "1"}, phases=True, deltaT=1e-3, adjustTimeStep=1, maxCo=0.2)
    a.pop("tiles"); b.pop("tiles")
    for k in a:
        assert np.array_equal(a[k], b[k]), k



def test_staged_kernel_stays_bit_identical_over_a_long_run():
    """300 steps on 48^3 cells (the bench's state and time step): any difference between the two kernels would be amplified"""
    mesh = q.PolyMesh.box(48, 48, 48)
    opt = dict(deltaT=0.1 / 48 / 1.3)
    a = run(mesh, 300, {"QGD_FTILE": "0"}, **opt)
    b = run(mesh, 300, {"QGD_FTILE": "1"}, **opt)
    assert b.pop("tiles")["facesPerTile"] == 128
    a.pop("tiles")
    for k in a:
        assert np.isfinite(a[k]).all() and np.array_equal(a[k], b[k]), k
