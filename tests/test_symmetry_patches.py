"""Constraint patches in the resident cases (missing #3).

OpenFOAM gives a field on a constraint patch the patch's own field type whatever the field file says: on
``symmetryPlane`` / ``symmetry`` patches U is reflected (basicSymmetry) and scalars are zero-gradient; cyclic and wedge
patches carry coupled / rotated fields the resident cases do not serve and therefore refuse.  The stencils know the same
list [extendedFaceStencilScalarGrad.C L90-101, GaussVolPointBase3D.C L783-794].

The CPU half asserts PROPERTIES the oracle cannot share a mistake about with the device: the patch velocity has no normal
component, no mass crosses the plane (boundary phiJm = 0), a closed box of symmetry planes conserves mass.  The GPU half
is parity of the HIP path against the oracle on the same cases (<= 1e-10), QGDFoam in both branches and QHDFoam.
"""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L

import cases
from oracle import OracleCase, OracleMesh, OracleQhdCase
from util import rel_err

G, E, SP, SY = L.PATCH_GENERIC, L.PATCH_EMPTY, L.PATCH_SYMMETRYPLANE, L.PATCH_SYMMETRY


def sym_mesh(kind):
    if kind == "box3d":      # yMin symmetryPlane, zMax symmetry, the rest ordinary patches
        return q.PolyMesh.box(6, 5, 4, patch_types=[G, G, SP, G, G, SY])
    if kind == "box3d_jitter":
        return q.PolyMesh.box(6, 5, 4, patch_types=[G, G, SP, G, G, SY]).jitter(0.12, seed=5)
    if kind == "closed3d":   # every wall a symmetry plane: nothing leaves the box
        return q.PolyMesh.box(5, 4, 3, patch_types=[SP, SP, SP, SP, SY, SY])
    if kind == "plane2d":    # forwardStep-like: symmetryPlane top and bottom, z empty
        return q.PolyMesh.box(8, 6, 1, hi=(1.0, 0.75, 0.1), patch_types=[G, G, SP, SP, E, E])
    raise KeyError(kind)


# boundary phiJm relative to the largest |phiJm| of the mesh.  On orthogonal cells the Gauss diamond of a wall face is mirror-symmetric
# and the flux vanishes to rounding; the reference's ghost point is the owner centre reflected through the face CENTRE
# [GaussVolPointBase3D.C L142-147], so on skewed cells a (tau-sized) remainder is the scheme's own
JM_TOL = {"box3d": 1e-13, "closed3d": 1e-13, "plane2d": 1e-13, "box3d_jitter": 2e-3}


def hostile_bcs(case):
    """what foamfile._bc used to hand over for constraint patches ("none"), and an outright contradiction: the patch type wins"""
    n = case.mesh.nPatches if hasattr(case.mesh, "nPatches") else 6
    for i in range(n):
        case.set_bc(i, U=("none", None), T=("none", None), p=("none", None)) if i % 2 == 0 else \
            case.set_bc(i, U=("fixedValue", (1.0, 2.0, 3.0)), T=("fixedValue", 7.0), p=("fixedValue", 9.0))


def initial_state(mesh):
    """the box state with a velocity towards the planes; on the one-cell-thick plane the pressure pulse sits in the plane"""
    C = mesh.array("C").reshape(-1, 3)
    U, T, p = cases.box_initial_fields(C)
    U[:, 1] += 0.2
    if mesh.nGeometricD == 3:
        U[:, 2] += 0.15
    else:
        p = 1.0 + 0.1 * np.exp(-((C[:, 0] - 0.5) ** 2 + (C[:, 1] - 0.4) ** 2) / 0.01)
    return U, T, p


def patch_faces(mesh, types):
    ps, pz, pt = mesh.array("patchStart"), mesh.array("patchSize"), mesh.array("patchType")
    out = []
    for i in range(mesh.nPatches):
        if int(pt[i]) in types:
            out.extend(range(int(ps[i]), int(ps[i]) + int(pz[i])))
    return np.asarray(out, dtype=np.int64)


def away_from_ordinary_patches(mesh, faces):
    """the faces that share no vertex with an ordinary patch.  The 2-D GaussVolPoint gradient of U interpolates U component by
    component [GaussVolPointBase.C L79-87] -- three scalar fields, which no point constraint touches -- so the vertex where a symmetry
    plane meets the inlet keeps a share of the inlet's normal velocity, and the plane's first face a tangential derivative of it"""
    fo, fp = mesh.array("faceOffsets"), mesh.array("facePoints")
    taken = set()
    for f in patch_faces(mesh, (G,)):
        taken.update(fp[fo[f]:fo[f + 1]].tolist())
    return np.asarray([f for f in faces if not taken.intersection(fp[fo[f]:fo[f + 1]].tolist())], dtype=np.int64)


def normal_velocity_and_mass_flux(case, mesh, inner_only=False):
    """(max |U_b . n|, max |phiJm| / max |phiJm| of the whole mesh) over the faces of the symmetry patches"""
    faces = patch_faces(mesh, (SP, SY))
    if inner_only:
        faces = away_from_ordinary_patches(mesh, faces)
        assert faces.size > 0
    Sf = mesh.array("Sf").reshape(-1, 3)
    n = Sf[faces] / np.linalg.norm(Sf[faces], axis=1)[:, None]
    Ub = case.field("U.boundary").reshape(-1, 3)[faces - mesh.nInternalFaces]
    case.updateFluxes()
    phi = case.field("phiJm")
    return float(np.abs((Ub * n).sum(axis=1)).max()), float(np.abs(phi[faces]).max() / max(np.abs(phi).max(), 1e-300))


@pytest.mark.parametrize("kind,scheme", [("box3d", "GaussVolPoint"), ("box3d", "reduced"), ("box3d_jitter", "GaussVolPoint"),
                                         ("plane2d", "leastSquares"), ("plane2d", "GaussVolPoint")])
def test_oracle_symmetry_patches_are_impermeable_whatever_the_caller_asks(kind, scheme):
    mesh = sym_mesh(kind)
    oc = OracleCase(OracleMesh(mesh.primitives()), q.default_options(stencil=scheme, deltaT=1e-3, mu=1e-3))
    oc.mesh.nPatches = mesh.nPatches
    ptypes = mesh.array("patchType")
    for i in range(mesh.nPatches):   # constraint patches: a contradiction; ordinary ones: zeroGradient
        if int(ptypes[i]) in (SP, SY):
            oc.set_bc(i, U=("zeroGradient", None), T=("fixedValue", 3.0), p=("fixedValue", 5.0))
    U, T, p = initial_state(mesh)   # drives flow at the planes
    oc.set_fields(U, T, p)
    for _ in range(3):
        un, jm = normal_velocity_and_mass_flux(oc, mesh, inner_only=(kind == "plane2d" and scheme == "GaussVolPoint"))
        assert un <= 1e-15, (kind, scheme, un)
        assert jm <= JM_TOL[kind], (kind, scheme, jm)
        oc.step(5)
    # scalars are zero-gradient there (the fixedValue request was overridden)
    faces = patch_faces(mesh, (SP, SY))
    own = mesh.array("owner")[faces]
    for name in ("p", "e"):
        assert np.array_equal(oc.field(name + ".boundary")[faces - mesh.nInternalFaces], oc.field(name)[own]), name


def test_oracle_closed_box_of_symmetry_planes_conserves_mass():
    mesh = sym_mesh("closed3d")
    oc = OracleCase(OracleMesh(mesh.primitives()), q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3))
    C = mesh.array("C").reshape(-1, 3)
    U, T, p = cases.box_initial_fields(C)
    U += 0.1
    oc.set_fields(U, T, p)
    V = mesh.array("V")
    m0 = float((oc.field("rho") * V).sum())
    oc.step(40)
    m1 = float((oc.field("rho") * V).sum())
    assert abs(m1 - m0) <= 1e-13 * m0, (m0, m1)
    assert np.abs(oc.field("rho") - 1.4).max() > 1e-4   # the state did move


def test_oracle_refuses_cyclic_and_wedge_cases():
    for t in (L.PATCH_CYCLIC, L.PATCH_WEDGE):
        mesh = q.PolyMesh.box(4, 3, 2, patch_types=[t, t, G, G, G, G])
        with pytest.raises(ValueError):
            OracleCase(OracleMesh(mesh.primitives()), q.default_options(stencil="reduced"))
    # an empty cyclic patch (size 0) is no obstacle: decomposePar leaves such patches behind


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path against the oracle on the same cases
# ---------------------------------------------------------------------------------------------------------------------
def _pair(kind, scheme, **opt):
    mesh = sym_mesh(kind)
    options = q.default_options(stencil=scheme, **opt)
    dev = q.Device(mesh)
    gc = q.QGDFoamCase(dev, options)
    oc = OracleCase(OracleMesh(mesh.primitives()), options)
    return mesh, dev, gc, oc, initial_state(mesh)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,scheme,opt", [
    ("box3d", "GaussVolPoint", dict(deltaT=1e-3, mu=1e-3)),
    ("box3d", "reduced", dict(deltaT=1e-3, mu=1e-3)),
    ("box3d_jitter", "GaussVolPoint", dict(deltaT=5e-4, mu=1e-3)),
    ("closed3d", "GaussVolPoint", dict(deltaT=1e-3, mu=2e-3)),
    ("plane2d", "leastSquares", dict(deltaT=1e-3)),
    ("plane2d", "GaussVolPoint", dict(deltaT=1e-3, mu=1e-3)),
    ("box3d", "GaussVolPoint", dict(deltaT=1e-3, mu=2e-3, implicitDiffusion=1, implicitTol=1e-14)),
    ("plane2d", "leastSquares", dict(deltaT=1e-3, mu=2e-3, implicitDiffusion=1, implicitTol=1e-14)),
])
def test_device_symmetry_patches_match_the_oracle_and_are_impermeable(kind, scheme, opt):
    mesh, dev, gc, oc, (U, T, p) = _pair(kind, scheme, **opt)
    ptypes = mesh.array("patchType")
    for case in (gc, oc):   # the caller's (wrong) request on the constraint patches is overridden on both sides
        for i in range(mesh.nPatches):
            if int(ptypes[i]) in (SP, SY):
                case.set_bc(i, U=("none", None), T=("none", None), p=("none", None))
        case.set_fields(U, T, p)
    un, jm = normal_velocity_and_mass_flux(gc, mesh, inner_only=(kind == "plane2d" and scheme == "GaussVolPoint"))
    assert un <= 1e-15 and jm <= JM_TOL[kind], (kind, scheme, un, jm)
    oc.updateFluxes()
    for name in ("phiJm", "phiJmU", "phiPi", "phiQ", "gradUf", "gradPf"):
        assert rel_err(gc.field(name), oc.field(name)) <= 1e-11, name
    for chunk in (1, 11):
        gc.step(chunk)
        oc.step(chunk)
        for name in ("rho", "U", "p", "e", "U.boundary", "p.boundary"):
            assert rel_err(gc.field(name), oc.field(name)) <= 1e-10, (kind, scheme, name, chunk)
    un, jm = normal_velocity_and_mass_flux(gc, mesh, inner_only=(kind == "plane2d" and scheme == "GaussVolPoint"))
    assert un <= 1e-15 and jm <= JM_TOL[kind], (kind, scheme, un, jm)
    gc.close(); dev.close()


@pytest.mark.gpu
def test_device_closed_box_of_symmetry_planes_conserves_mass():
    mesh, dev, gc, oc, (U, T, p) = _pair("closed3d", "GaussVolPoint", deltaT=1e-3, mu=1e-3)
    gc.set_fields(U + 0.1, T, p)
    V = mesh.array("V")
    m0 = float((gc.field("rho") * V).sum())
    gc.step(40)
    m1 = float((gc.field("rho") * V).sum())
    assert abs(m1 - m0) <= 1e-13 * m0, (m0, m1)
    gc.close(); dev.close()


@pytest.mark.gpu
def test_device_refuses_cyclic_and_wedge_cases_and_non_planar_symmetry_planes():
    from qgdsolver_amd.qhdfoam import QHDFoamCase, qhd_options
    for t, word in ((L.PATCH_CYCLIC, "cyclic"), (L.PATCH_WEDGE, "wedge")):
        mesh = q.PolyMesh.box(4, 3, 2, patch_types=[t, t, G, G, G, G])
        dev = q.Device(mesh)
        with pytest.raises(L.QgdError, match=word) as e:
            q.QGDFoamCase(dev, q.default_options(stencil="reduced"))
        assert e.value.code == L.ERR_NOT_IMPLEMENTED
        with pytest.raises(L.QgdError, match=word):
            QHDFoamCase(dev, qhd_options(stencil="reduced"))
        dev.close()
    # a symmetryPlane whose faces do not share one normal is fatal in OpenFOAM (symmetryPlanePolyPatch::calcGeometry)
    mesh = q.PolyMesh.box(4, 3, 2, patch_types=[G, G, SP, G, G, G])
    pts = mesh.array("points").reshape(-1, 3).copy()
    low = np.where((pts[:, 1] == 0.0) & (pts[:, 0] > 0.4) & (pts[:, 0] < 0.6))[0]
    pts[low, 1] -= 0.05
    bent = q.PolyMesh.from_arrays(pts.reshape(-1), mesh.array("faceOffsets"), mesh.array("facePoints"), mesh.array("owner"),
                                  mesh.array("neighbour"), mesh.nCells, mesh.array("patchStart"), mesh.array("patchSize"),
                                  mesh.array("patchType"))
    dev = q.Device(bent)
    with pytest.raises(L.QgdError, match="not planar"):
        q.QGDFoamCase(dev, q.default_options(stencil="reduced"))
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("implicit", [0, 1])
def test_qhd_case_on_symmetry_planes(implicit):
    from qgdsolver_amd.qhdfoam import QHDFoamCase, qhd_options
    mesh = sym_mesh("box3d")
    # gravity ALONG the planes: a body force with a component normal to a symmetry plane drives phiwo through it (p is zero-gradient there,
    # nothing balances beta*T*g.n -- the reference's own behaviour, what its qhdFlux patch exists for on walls)
    opt = qhd_options(stencil="GaussVolPoint", deltaT=2e-3, mu=1e-2, Pr=0.7, beta=3e-3, g=(-9.81, 0.0, 0.0), tauModel="constTau", Tau=1e-3,
                      implicitDiffusion=implicit, implicitTol=1e-13, pTol=1e-13)
    dev = q.Device(mesh)
    gc = QHDFoamCase(dev, opt)
    oc = OracleQhdCase(OracleMesh(mesh.primitives()), opt)
    C = mesh.array("C").reshape(-1, 3)
    U = 0.05 * np.stack([np.sin(2 * np.pi * C[:, 0]) * np.cos(np.pi * C[:, 1]), 0.5 + 0 * C[:, 0], 0.3 * np.cos(np.pi * C[:, 2])], axis=1)
    T = 300.0 + 5.0 * C[:, 0]
    p = np.zeros(mesh.nCells)
    ptypes = mesh.array("patchType")
    for case in (gc, oc):
        for i in range(mesh.nPatches):
            if int(ptypes[i]) in (SP, SY):
                case.set_bc(i, U=("fixedValue", (1.0, 1.0, 1.0)), T=("fixedValue", 1.0), p=("fixedValue", 2.0))   # overridden by the patch type
            else:
                case.set_bc(i, U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("zeroGradient", None))
        case.set_fields(U, T, p)
    gc.step(8)
    oc.step(8)
    for name in ("U", "T", "p", "phi"):
        assert rel_err(gc.field(name), oc.field(name)) <= 1e-9, (implicit, name)
    faces = patch_faces(mesh, (SP, SY))
    Sf = mesh.array("Sf").reshape(-1, 3)
    n = Sf[faces] / np.linalg.norm(Sf[faces], axis=1)[:, None]
    Ub = gc.field("U.boundary").reshape(-1, 3)[faces - mesh.nInternalFaces]
    assert np.abs((Ub * n).sum(axis=1)).max() <= 1e-16
    assert np.abs(gc.field("phi")[faces]).max() <= 1e-13 * max(np.abs(gc.field("phi")).max(), 1e-300)
    gc.close(); dev.close()


def _sharding_bcs(case):
    """patches 0, 1, 4 ordinary (inlet / outlet / a qgdFlux wall); 2 (symmetryPlane), 3 (wall), 5 (symmetry)"""
    case.set_bc(0, U=("fixedValue", (0.1, 0.0, 0.0)), T=("fixedValue", 1.05), p=("zeroGradient", None))
    case.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
    case.set_bc(3, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))
    case.set_bc(4, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))


@pytest.mark.gpu
@pytest.mark.parametrize("scheme,world,opt", [
    ("GaussVolPoint", 3, dict(deltaT=5e-4, mu=1e-3, fluxSchemeU=1, fluxSchemeH=1)),
    ("GaussVolPoint", 4, dict(deltaT=5e-4, mu=1e-3, termStencils={"grad(p)": "reduced"})),
    ("reduced", 2, dict(deltaT=5e-4, mu=1e-3)),
])
def test_sharded_symmetry_upwind_and_mixed_stencil_cases_match_the_unsharded_run(scheme, world, opt):
    """this round's case features on cell-range shards (extractShard: a shard inherits the symmetry plane's patch normal; its point
    constraints come out of the faces around each vertex, which a vertex-connected ghost layer holds completely): several shards
    resident on one GPU, messages through device buffers, against the unsharded device run (1e-12) and the oracle (1e-10)"""
    from test_partition import random_perm, run_oracle
    from test_partition_gpu import run_device, run_sharded_device
    g = sym_mesh("box3d_jitter")
    g.renumber(random_perm(g.nCells, 17))
    U, T, p = initial_state(g)
    steps = 8
    ref = run_oracle(g, scheme, _sharding_bcs, U, T, p, steps, **opt)
    one = run_device(g, scheme, _sharding_bcs, U, T, p, steps, **opt)
    got = run_sharded_device(g, world, scheme, _sharding_bcs, U, T, p, steps, overlapped=(world == 3), **opt)
    for f in ref:
        scale = np.abs(ref[f]).max()
        assert np.abs(got[f] - one[f]).max() <= 1e-12 * scale, (scheme, f, "sharded vs unsharded device")
        assert np.abs(got[f] - ref[f]).max() <= 1e-10 * scale, (scheme, f, "sharded device vs oracle")
