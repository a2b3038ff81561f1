"""Helpers shared by the parity tests."""
import numpy as np

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L

from oracle import OracleCase, OracleMesh


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): relative to the field's own scale (the 1e-10 bar of north_star)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300) if b.size else 1.0
    return float(np.abs(a - b).max() / scale) if b.size else 0.0


def make_mesh(kind):
    """Named test meshes.  Returns a PolyMesh."""
    G, E = L.PATCH_GENERIC, L.PATCH_EMPTY
    if kind == "box654":
        return q.PolyMesh.box(6, 5, 4)
    if kind == "box654_jitter":
        return q.PolyMesh.box(6, 5, 4).jitter(0.15, seed=2024)
    if kind == "box654_tri":
        return q.PolyMesh.box(6, 5, 4).jitter(0.1, seed=7).split_quads(3)
    if kind == "box654_poly":  # pentagon / hexagon faces (mid-edge vertices) + triangles + jitter
        return q.PolyMesh.box(6, 5, 4).jitter(0.1, seed=3).split_quads(5).split_edges(4)
    if kind == "plane2d":  # 8 x 7 x 1, z empty
        return q.PolyMesh.box(8, 7, 1, hi=(1.0, 0.875, 0.1), patch_types=[G, G, G, G, E, E])
    if kind == "plane2d_jitter":
        return q.PolyMesh.box(8, 7, 1, hi=(1.0, 0.875, 0.1), patch_types=[G, G, G, G, E, E]).jitter(0.15, seed=11)
    if kind == "plane2d_y":  # empty direction y
        return q.PolyMesh.box(8, 1, 6, hi=(1.0, 0.1, 0.75), patch_types=[G, G, E, E, G, G])
    if kind == "line1d":  # 20 x 1 x 1
        return q.PolyMesh.box(20, 1, 1, hi=(1.0, 0.05, 0.05), patch_types=[G, G, E, E, E, E])
    if kind == "step2d":
        return q.PolyMesh.forward_step(30, 10, 6, 2, lx=3.0, ly=1.0, lz=0.1)
    if kind == "box_sym":  # symmetryPlane on yMin: constraint patch for leastSquares / GaussVolPoint rules
        return q.PolyMesh.box(6, 5, 1, hi=(1.0, 1.0, 0.1), patch_types=[G, G, L.PATCH_SYMMETRYPLANE, G, E, E])
    raise KeyError(kind)


def oracle_mesh_of(mesh):
    return OracleMesh(mesh.primitives())


def expects_fused(mesh, options):
    """whether a case with these options is one the fused one-launch step (fusedFaceCellKernel, the kernel bench.py times) serves: a 3-D mesh,
    one GaussVolPoint stencil, explicit branch, fixed deltaT (qgd_capi.cpp qgd_case_create)"""
    return (mesh.nGeometricD == 3 and options.stencil == L.FVSC_GAUSSVOLPOINT and not options.implicitDiffusion and
            not options.adjustTimeStep and not any(options.termStencil))


def expects_fused_adjust(mesh, options):
    """... and under Courant-number control the same blocks run up to their flux sums, a cell kernel advances once deltaT is known"""
    return (mesh.nGeometricD == 3 and options.stencil == L.FVSC_GAUSSVOLPOINT and not options.implicitDiffusion and
            bool(options.adjustTimeStep) and not any(options.termStencil))


def device_pair_arms(mesh, options):
    """the two ways a device can step such a case, for tests that pin BOTH to the oracle on purpose: ("fused", Device with the block tables
    whatever the blocks look like) and ("kernels", Device without them); a case the fused step does not serve has the second arm only"""
    arms = [("kernels", False)]
    if expects_fused(mesh, options):
        arms.insert(0, ("fused", "any"))
    return arms


def assert_path(case, arm, tag=None):
    """the arm under test is the path that runs -- asserted, not left to a default or to the block-size heuristic"""
    info = case.fused_info()
    if arm == "fusedAdjust":
        assert info["fusedAdjust"] and not info["fused"] and info["blocks"] >= 1, (tag, arm, info)
        return
    assert info["fused"] == (arm == "fused") and not info["fusedAdjust"], (tag, arm, info)
    if arm == "fused":
        assert info["blocks"] >= 1 and info["facesComputed"] >= case.dev.mesh.nInternalFaces, (tag, info)
