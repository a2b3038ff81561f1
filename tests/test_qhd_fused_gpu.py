"""QHDFoam's U and T equations on the cell blocks of QGDFoam's one-launch step (QGD_QHD_FUSED=1, off by default; qgd_qhd.hip qhdFusedAdvanceKernel: the vertex
values of p, face pass 2 and the explicit Euler update [QHDUEqn.H L36-84, QHDTEqn.H L65-91] as one launch) against the separate kernels of the
same library (a device without block tables) and against the oracle."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
from qgdsolver_amd.synthetic import c5_mesh

from oracle import OracleQhdCase
from test_qhd_case import cavity_bcs, initial, options
from util import oracle_mesh_of

FIELDS = ("U", "T", "p", "phi", "U.boundary", "T.boundary", "p.boundary")


def _meshes():
    scr = q.PolyMesh.box(12, 10, 8)
    scr.renumber(np.random.default_rng(5).permutation(scr.nCells).astype(np.int32))
    return (("hex 37x11x5", q.PolyMesh.box(37, 11, 5)), ("hex 20^3 jittered", q.PolyMesh.box(20, 20, 20).jitter(0.15, seed=4)),
            ("triangles + polygons, Morton order", c5_mesh(16, 8 ** 3, poly=True)), ("scrambled labels", scr))


def _run(mesh, tables, steps, **kw):
    dev = q.Device(mesh, fused_tables=tables)
    c = qhdfoam.QHDFoamCase(dev, options(deltaT=1e-3, pTol=1e-12, **kw))
    cavity_bcs(c, mesh)
    U, T, p = initial(mesh)
    U = U + 1e-2 * np.random.default_rng(3).standard_normal(U.shape)
    c.set_fields(U, T, p)
    fi = c.fused_info()
    c.step(steps)
    out = {k: c.field(k) for k in FIELDS}
    info = c.info()
    c.close(); dev.close()
    return out, fi, info


@pytest.mark.gpu
@pytest.mark.parametrize("upwind", [0, 1])
def test_u_and_t_equations_on_the_blocks_match_the_separate_kernels(upwind, monkeypatch):
    monkeypatch.setenv("QGD_QHD_FUSED", "1")
    kw = dict(fluxSchemeU=q._lib.FLUX_UPWIND, fluxSchemeT=q._lib.FLUX_UPWIND) if upwind else {}
    for tag, mesh in _meshes():
        sep, fs, _ = _run(mesh, False, 6, **kw)
        fus, ff, info = _run(mesh, "any", 6, **kw)
        assert not fs["fusedAdvance"] and ff["fusedAdvance"] and ff["blocks"] > 0 and 0 < ff["ldsAdvance"] <= 65536, (tag, fs, ff)
        for k in FIELDS:
            scale = max(np.abs(sep[k]).max(), 1e-300)
            assert np.isfinite(fus[k]).all() and (k == "U.boundary" or np.abs(sep[k]).max() > 0)
            assert np.abs(fus[k] - sep[k]).max() <= 1e-11 * scale, (tag, k, np.abs(fus[k] - sep[k]).max() / scale)
        assert info["steps"] == 6


@pytest.mark.gpu
def test_blocks_are_not_used_where_they_do_not_apply(monkeypatch):
    monkeypatch.setenv("QGD_QHD_FUSED", "1")
    mesh = q.PolyMesh.box(16, 12, 8)
    for kw, what in ((dict(implicitDiffusion=1), "implicitDiffusion"), (dict(stencil="reduced"), "another stencil")):
        dev = q.Device(mesh, fused_tables="any")
        opt = options(deltaT=1e-3, **{k: v for k, v in kw.items() if k != "stencil"}) if "stencil" not in kw else options(kw["stencil"], deltaT=1e-3)
        c = qhdfoam.QHDFoamCase(dev, opt)
        assert not c.fused_info()["fusedAdvance"], what
        c.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["hex", "poly"])
def test_block_fused_case_matches_the_oracle(kind, monkeypatch):
    monkeypatch.setenv("QGD_QHD_FUSED", "1")
    mesh = q.PolyMesh.box(9, 8, 7).jitter(0.1, seed=2) if kind == "hex" else c5_mesh(8, 4 ** 3, poly=True)
    om = oracle_mesh_of(mesh)
    opt = options(deltaT=1e-3)
    dev = q.Device(mesh, fused_tables="any")
    gc, oc = qhdfoam.QHDFoamCase(dev, opt), OracleQhdCase(om, opt)
    U, T, p = initial(mesh)
    U = U + 1e-2 * np.random.default_rng(3).standard_normal(U.shape)
    for c in (gc, oc):
        cavity_bcs(c, mesh)
        c.set_fields(U, T, p)
    assert gc.fused_info()["fusedAdvance"]
    gc.step(10); oc.step(10)
    for f in FIELDS:
        ref = oc.field(f)
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(gc.field(f) - ref).max() <= 1e-9 * scale, (kind, f, np.abs(gc.field(f) - ref).max() / scale)
    assert np.abs(oc.field("U")).max() > 1e-3
    gc.close(); dev.close()
