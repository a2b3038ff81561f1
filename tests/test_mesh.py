"""CPU: mesh generator connectivity (bit-for-bit against an independent numpy construction), OpenFOAM ordering
rules and geometry identities; the product's geometry against the oracle's own implementation."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L

from util import make_mesh, oracle_mesh_of

HEX_FACES = [(0, 4, 7, 3), (1, 2, 6, 5), (0, 1, 5, 4), (3, 7, 6, 2), (0, 3, 2, 1), (4, 5, 6, 7)]


def numpy_box(nx, ny, nz):
    """Independent blockMesh-style box: cell i + nx (j + ny k); internal faces by (owner, neighbour)."""
    P = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
    def verts(i, j, k):
        return [P(i, j, k), P(i + 1, j, k), P(i + 1, j + 1, k), P(i, j + 1, k),
                P(i, j, k + 1), P(i + 1, j, k + 1), P(i + 1, j + 1, k + 1), P(i, j + 1, k + 1)]
    cell = lambda i, j, k: i + nx * (j + ny * k)
    faces, own, nei = [], [], []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                v = verts(i, j, k)
                for ok, hf, other in ((i < nx - 1, 1, (i + 1, j, k)), (j < ny - 1, 3, (i, j + 1, k)), (k < nz - 1, 5, (i, j, k + 1))):
                    if ok:
                        faces.append([v[q_] for q_ in HEX_FACES[hf]]); own.append(cell(i, j, k)); nei.append(cell(*other))
    nif = len(faces)
    starts, sizes = [], []
    def patch(gen, hf):
        starts.append(len(faces))
        for (i, j, k) in gen:
            v = verts(i, j, k)
            faces.append([v[q_] for q_ in HEX_FACES[hf]]); own.append(cell(i, j, k))
        sizes.append(len(faces) - starts[-1])
    patch(((0, j, k) for k in range(nz) for j in range(ny)), 0)
    patch(((nx - 1, j, k) for k in range(nz) for j in range(ny)), 1)
    patch(((i, 0, k) for i in range(nx) for k in range(nz)), 2)
    patch(((i, ny - 1, k) for i in range(nx) for k in range(nz)), 3)
    patch(((i, j, 0) for i in range(nx) for j in range(ny)), 4)
    patch(((i, j, nz - 1) for i in range(nx) for j in range(ny)), 5)
    return np.array(faces, np.int32), np.array(own, np.int32), np.array(nei, np.int32), nif, starts, sizes


@pytest.mark.parametrize("dims", [(4, 3, 2), (1, 1, 1), (5, 1, 1), (3, 4, 5)])
def test_box_connectivity_bit_exact(dims):
    nx, ny, nz = dims
    m = q.PolyMesh.box(nx, ny, nz)
    faces, own, nei, nif, starts, sizes = numpy_box(nx, ny, nz)
    assert m.nCells == nx * ny * nz and m.nInternalFaces == nif and m.nFaces == len(faces)
    assert np.array_equal(m.array("facePoints").reshape(-1, 4), faces)
    assert np.array_equal(m.array("owner"), own)
    assert np.array_equal(m.array("neighbour"), nei)
    assert list(m.array("patchStart")) == starts and list(m.array("patchSize")) == sizes
    assert np.array_equal(m.array("faceOffsets"), 4 * np.arange(m.nFaces + 1))


@pytest.mark.parametrize("kind", ["box654", "box654_jitter", "box654_tri", "plane2d", "step2d", "line1d"])
def test_ordering_and_geometry_identities(kind):
    m = make_mesh(kind)
    own, nei = m.array("owner"), m.array("neighbour")
    nif = m.nInternalFaces
    # upper-triangular order: owner < neighbour, faces sorted by (owner, neighbour)
    assert (own[:nif] < nei).all()
    key = own[:nif].astype(np.int64) * m.nCells + nei
    assert (np.diff(key) >= 0).all()
    # closed cells: sum of outward area vectors vanishes; volumes positive and add up
    Sf = m.array("Sf").reshape(-1, 3)
    acc = np.zeros((m.nCells, 3))
    np.add.at(acc, own, Sf)
    np.subtract.at(acc, nei, Sf[:nif])
    assert np.abs(acc).max() < 1e-14
    V = m.array("V")
    assert (V > 0).all()
    # Gauss: V = 1/3 sum Cf.Sf
    Cf = m.array("Cf").reshape(-1, 3)
    g = np.zeros(m.nCells)
    np.add.at(g, own, (Cf * Sf).sum(1))
    np.subtract.at(g, nei, (Cf[:nif] * Sf[:nif]).sum(1))
    assert np.allclose(g / 3.0, V, rtol=1e-12, atol=1e-16)
    w = m.array("weights")
    assert ((w[:nif] > 0) & (w[:nif] < 1)).all() and (w[nif:] == 1).all()


def test_box_volume_and_directions():
    m = q.PolyMesh.box(7, 5, 3, lo=(0, 0, 0), hi=(2.0, 1.0, 0.5))
    assert abs(m.array("V").sum() - 1.0) < 1e-14
    assert m.nGeometricD == 3
    assert make_mesh("plane2d").nGeometricD == 2
    assert make_mesh("line1d").nGeometricD == 1
    assert oracle_mesh_of(make_mesh("plane2d_y")).info()["geometricD"] == [1, -1, 1]


@pytest.mark.parametrize("kind", ["box654_jitter", "box654_tri", "step2d"])
def test_geometry_matches_oracle_implementation(kind):
    """Two independent implementations of OpenFOAM's face/cell decomposition agree."""
    m = make_mesh(kind)
    om = oracle_mesh_of(m)
    for name in ("Sf", "magSf", "Cf", "C", "V", "weights", "deltaCoeffs", "nonOrthDeltaCoeffs"):
        a, b = m.array(name), om.array(name)
        assert np.abs(a - b).max() <= 1e-15 * max(1.0, np.abs(b).max()), name


def test_slab_is_a_window_of_the_global_box():
    """A k-slab shard carries the same points as the global mesh bit for bit; owned cells keep bit-identical
    centres/volumes, the ghost planes agree to rounding (their cut faces change from internal to boundary, which
    changes the summation order of the cell-centre accumulation)."""
    g = q.PolyMesh.box(5, 4, 9)
    s = q.PolyMesh.box(5, 4, 9, k_range=(2, 7))
    plane = 5 * 4
    pplane = 6 * 5
    assert np.array_equal(s.array("points"), g.array("points")[3 * pplane * 2: 3 * pplane * 8])
    Cs, Cg = s.array("C").reshape(-1, 3), g.array("C").reshape(-1, 3)[plane * 2: plane * 7]
    assert np.array_equal(Cs[plane:-plane], Cg[plane:-plane])
    assert np.abs(Cs - Cg).max() < 1e-15
    assert np.array_equal(s.array("V")[plane:-plane], g.array("V")[plane * 3: plane * 6])
    assert np.abs(s.array("V") - g.array("V")[plane * 2: plane * 7]).max() < 1e-17
    assert list(s.array("patchType")[4:]) == [L.PATCH_HALO, L.PATCH_HALO]
    assert np.array_equal(s.array("haloGhost0"), np.arange(plane))
    assert np.array_equal(s.array("haloSend0"), np.arange(plane, 2 * plane))
    assert np.array_equal(s.array("haloGhost1"), np.arange(4 * plane, 5 * plane))
    assert np.array_equal(s.array("haloSend1"), np.arange(3 * plane, 4 * plane))


def test_mesh_create_rejects_bad_order():
    p = q.PolyMesh.box(2, 2, 1).primitives()
    bad = dict(p)
    ne = p["neighbour"].copy(); ow = p["owner"].copy()
    ow[0], ne[0] = ne[0], ow[0]
    with pytest.raises(q.QgdError):
        q.PolyMesh.from_arrays(p["points"], p["faceOffsets"], p["facePoints"], ow, ne, p["nCells"], p["patchStart"],
                               p["patchSize"], p["patchType"])
    ok = q.PolyMesh.from_arrays(p["points"], p["faceOffsets"], p["facePoints"], p["owner"], p["neighbour"], p["nCells"],
                                p["patchStart"], p["patchSize"], p["patchType"])
    assert ok.nCells == 4
