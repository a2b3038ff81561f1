"""The faceSet "degenerateStencilFaces" of the leastSquares stencil [leastSquaresStencil.C L58-133]: user-listed internal faces take
the nf * snGrad fallback of extendedFaceStencilScalarGrad.C L76-83 whatever their weights say.

CPU: the list travels with the mesh (renumbering, sharding, the polyMesh/sets reader) and the oracle applies it; GPU: the device
against the oracle, and the listed faces against the reduced stencil."""
import os

import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import foamfile, fvsc

import cases
from util import make_mesh, oracle_mesh_of, rel_err


def listed_faces(mesh):
    return np.arange(0, mesh.nInternalFaces, 5, dtype=np.int32)


def test_oracle_uses_the_reduced_form_on_the_listed_faces():
    mesh = make_mesh("plane2d_jitter")
    faces = listed_faces(mesh)
    cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, 1, 3)
    plain = oracle_mesh_of(mesh)
    marked = oracle_mesh_of(mesh)
    marked.set_degenerate_faces(faces)
    _, g0 = plain.fvsc("leastSquares", "grad_s", cell, bnd)
    _, g1 = marked.fvsc("leastSquares", "grad_s", cell, bnd)
    _, red = plain.fvsc("reduced", "grad_s", cell, bnd)
    others = np.setdiff1d(np.arange(mesh.nInternalFaces), faces)
    assert np.array_equal(g1[others], g0[others])
    assert np.abs(g1[faces] - red[faces]).max() <= 1e-14 * np.abs(red).max()
    assert np.abs(g0[faces] - red[faces]).max() > 1e-3 * np.abs(red).max()      # it does change something


def test_the_list_follows_renumbering_sharding_and_the_case_files(tmp_path):
    mesh = make_mesh("plane2d_jitter")
    faces = listed_faces(mesh)
    mesh.set_degenerate_faces(faces)
    assert np.array_equal(mesh.array("degenerateFaces"), faces)
    own, nei = mesh.array("owner").copy(), mesh.array("neighbour").copy()
    pairs = {(int(own[f]), int(nei[f])) for f in faces}
    perm = np.random.default_rng(2).permutation(mesh.nCells).astype(np.int32)
    mesh.renumber(perm)
    o2, n2 = mesh.array("owner"), mesh.array("neighbour")
    inv = np.argsort(perm)
    got = {tuple(sorted((int(inv[o2[f]]), int(inv[n2[f]])))) for f in mesh.array("degenerateFaces")}
    assert got == {tuple(sorted(p)) for p in pairs}
    shard = mesh.shard(2, 1)
    fg = shard.array("faceGlobal")
    gl = np.where(fg >= 0, fg, -1 - fg)
    assert set(gl[shard.array("degenerateFaces")]) == set(mesh.array("degenerateFaces")) & set(gl[:shard.nInternalFaces].tolist() + gl[shard.nInternalFaces:].tolist())
    # constant/polyMesh/sets/degenerateStencilFaces
    m2 = make_mesh("plane2d_jitter")
    poly = os.path.join(tmp_path, "constant", "polyMesh")
    foamfile.write_polymesh(m2, poly)
    os.makedirs(os.path.join(poly, "sets"))
    with open(os.path.join(poly, "sets", "degenerateStencilFaces"), "w") as f:
        f.write("FoamFile\n{\n    version 2.0;\n    format ascii;\n    class faceSet;\n    location \"constant/polyMesh/sets\";\n    object degenerateStencilFaces;\n}\n\n")
        f.write(f"{faces.size}\n(\n" + "\n".join(str(int(x)) for x in faces) + "\n)\n")
    m3 = foamfile.read_polymesh(poly)
    assert np.array_equal(m3.array("degenerateFaces"), faces)
    with pytest.raises(q.QgdError):
        m3.set_degenerate_faces([m3.nFaces])


@pytest.mark.gpu
def test_device_applies_the_listed_faces_like_the_oracle():
    mesh = make_mesh("plane2d_jitter")
    faces = listed_faces(mesh)
    mesh.set_degenerate_faces(faces)
    om = oracle_mesh_of(mesh)
    om.set_degenerate_faces(faces)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "leastSquares", "grad(r)": "reduced"}})
    for op, nc in (("grad_s", 1), ("grad_v", 3), ("div_v", 3), ("div_t", 9)):
        cell, bnd = cases.random_fields(mesh.nCells, mesh.nBoundaryFaces, nc, 7 + nc)
        _, ref = om.fvsc("leastSquares", op, cell, bnd)
        vf = q.volField("f", cell, bnd)
        got = fvsc.grad(dev, vf) if op.startswith("grad") else fvsc.div(dev, vf)
        assert rel_err(got, ref) <= 1e-12, op
        if op == "grad_s":
            red = fvsc.grad(dev, q.volField("r", cell, bnd))
            assert np.abs(got[faces] - red[faces]).max() <= 1e-14 * np.abs(red).max()
    # and through the fused face kernel of the case
    from oracle import OracleCase
    from test_case_parity_gpu import empty_z_bcs, plane_init
    opt = q.default_options(stencil="leastSquares", deltaT=5e-4, mu=1e-3)
    gc, oc = q.QGDFoamCase(dev, opt), OracleCase(om, opt)
    U, T, p = plane_init(mesh.array("C").reshape(-1, 3))
    for c in (gc, oc):
        empty_z_bcs(c)
        c.set_fields(U, T, p)
        c.step(10)
    for f in ("rho", "U", "p", "e"):
        assert rel_err(gc.field(f), oc.field(f)) <= 1e-10, f
    gc.close(); dev.close()
