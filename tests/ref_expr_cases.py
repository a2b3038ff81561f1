"""Single-face meshes for the golden vectors of tests/golden/make_ref_expr.py (numbers evaluated from the reference's own
listing text): one internal face between cells whose centres are prescribed through set_geometry."""
import numpy as np

GENERIC, EMPTY = 0, 1


def load(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"ref_expr_{name}.npz"))


def unit(k):
    e = np.zeros(3)
    e[k] = 1.0
    return e


def two_cell_mesh(pts, nv, Sf, Cf, C, empty_normals=()):
    """primitives + geometry of: internal face 0 (vertices 0..nv-1) between cells 0 and 1; one `empty` boundary face per
    entry of empty_normals (owned by cell 0, reusing the first three vertices; its normal only sets the mesh's empty directions)"""
    pts = np.asarray(pts, float)[:nv]
    faces = [list(range(nv))] + [[0, 1, 2] for _ in empty_normals]
    owner = [0] * len(faces)
    prim = dict(points=pts.reshape(-1), faceOffsets=np.cumsum([0] + [len(f) for f in faces]).astype(np.int32),
                facePoints=np.concatenate(faces).astype(np.int32), owner=np.array(owner, np.int32), neighbour=np.array([1], np.int32),
                nCells=2, patchStart=np.array([1] if empty_normals else [], np.int32),
                patchSize=np.array([len(empty_normals)] if empty_normals else [], np.int32),
                patchType=np.array([EMPTY] if empty_normals else [], np.int32))
    S = [np.asarray(Sf, float)] + [np.asarray(n, float) for n in empty_normals]
    c = [np.asarray(Cf, float)] + [np.asarray(C[0], float) + 0.1 * np.asarray(n, float) for n in empty_normals]
    geom = dict(Sf=np.array(S), Cf=np.array(c), C=np.asarray(C, float), V=np.ones(2))
    return prim, geom


def boundary_face_mesh(pts, nv, Sf, Cf, C, back_axis=0, empty_normals=(), patch_type=GENERIC):
    """cell 0 (centre C) with ONE boundary face on a generic patch: face 1, vertices 0..nv-1, area vector / centre prescribed; an
    internal quad (face 0) a unit length behind the cell along -e_back_axis connects it to cell 1 and shares no vertex with the
    boundary face, so the boundary face's vertices are boundary points of that face alone.  One `empty` face per entry of
    empty_normals (owned by cell 0, on the back quad's vertices; its normal only sets the mesh's empty directions)."""
    pts = np.asarray(pts, float)[:nv]
    C = np.asarray(C, float)
    back = C - unit(back_axis)
    u, v = unit((back_axis + 1) % 3), unit((back_axis + 2) % 3)
    quad = [back + a * u + b * v for a, b in ((-0.5, -0.5), (-0.5, 0.5), (0.5, 0.5), (0.5, -0.5))]
    faces = [list(range(nv, nv + 4)), list(range(nv))] + [[nv, nv + 1, nv + 2] for _ in empty_normals]
    n_e = len(empty_normals)
    prim = dict(points=np.concatenate([pts, np.array(quad)]).reshape(-1), faceOffsets=np.cumsum([0] + [len(f) for f in faces]).astype(np.int32),
                facePoints=np.concatenate(faces).astype(np.int32), owner=np.zeros(len(faces), np.int32), neighbour=np.array([1], np.int32),
                nCells=2, patchStart=np.array([1, 2][:1 + (n_e > 0)], np.int32), patchSize=np.array([1, n_e][:1 + (n_e > 0)], np.int32),
                patchType=np.array([patch_type, EMPTY][:1 + (n_e > 0)], np.int32))
    geom = dict(Sf=np.array([-unit(back_axis), np.asarray(Sf, float)] + [np.asarray(n, float) for n in empty_normals]),
                Cf=np.array([back, np.asarray(Cf, float)] + [C + 0.1 * np.asarray(n, float) for n in empty_normals]),
                C=np.array([C, C - 2.0 * unit(back_axis)]), V=np.ones(2))
    return prim, geom


def lsq_mesh(n, Cf, centres, one_d):
    """internal face 0 between cells 0 and 1 (a quad normal to x around Cf); cells 2..n-1 hang on vertex 0 of that face through
    one generic boundary triangle each, so that the face's point-neighbour stencil is cells 0..n-1 in this order
    [extendedFaceStencilFindNeighbours.C L55-80]; empty faces make the mesh 2-D (z) or 1-D (y and z)"""
    Cf = np.asarray(Cf, float)
    cen = np.asarray(centres, float)[:n]
    h = 0.1 * np.abs(cen[0] - Cf).max()
    quad = [Cf + h * np.array([0, -1, -1]), Cf + h * np.array([0, 1, -1]), Cf + h * np.array([0, 1, 1]), Cf + h * np.array([0, -1, 1])]
    pts = list(quad)
    faces, owner, S, c = [[0, 1, 2, 3]], [0], [np.array([4 * h * h, 0, 0])], [Cf]
    for cell in range(2, n):
        a = len(pts)
        pts += [cen[cell] + h * np.array([0, 1, 0]), cen[cell] + h * np.array([0, 0, 1])]
        faces.append([0, a, a + 1]); owner.append(cell)
        S.append(np.array([0.0, h * h, 0.0])); c.append(cen[cell] + h * np.array([0, 1, 0]))
    n_generic = n - 2
    empties = [unit(2)] + ([unit(1)] if one_d else [])
    for e in empties:
        faces.append([0, 1, 2]); owner.append(0); S.append(h * h * e); c.append(cen[0] + h * e)
    prim = dict(points=np.array(pts).reshape(-1), faceOffsets=np.cumsum([0] + [len(f) for f in faces]).astype(np.int32),
                facePoints=np.concatenate(faces).astype(np.int32), owner=np.array(owner, np.int32), neighbour=np.array([1], np.int32),
                nCells=n, patchStart=np.array([1, 1 + n_generic], np.int32), patchSize=np.array([n_generic, len(empties)], np.int32),
                patchType=np.array([GENERIC, EMPTY], np.int32))
    geom = dict(Sf=np.array(S), Cf=np.array(c), C=cen, V=np.ones(n))
    return prim, geom
