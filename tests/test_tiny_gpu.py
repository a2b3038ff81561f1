"""Degenerate sizes: one cell (no internal face at all), two cells (one internal face), a single row -- through the whole
path (fvsc operators, flux assembly, steps) against the oracle."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L
from oracle import OracleCase
from util import oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu
G, E = L.PATCH_GENERIC, L.PATCH_EMPTY


@pytest.mark.parametrize("dims,ptypes,stencil", [
    ((1, 1, 1), None, "GaussVolPoint"),
    ((1, 1, 1), None, "reduced"),
    ((2, 1, 1), None, "GaussVolPoint"),
    ((3, 1, 1), [G, G, E, E, E, E], "GaussVolPoint"),
    ((2, 2, 1), [G, G, G, G, E, E], "leastSquares"),
    ((2, 2, 1), [G, G, G, G, E, E], "GaussVolPoint"),
])
def test_tiny_meshes(dims, ptypes, stencil):
    mesh = q.PolyMesh.box(*dims, patch_types=ptypes)
    assert mesh.nInternalFaces == (dims[0] - 1) * dims[1] * dims[2] + dims[0] * (dims[1] - 1) * dims[2] + dims[0] * dims[1] * (dims[2] - 1)
    n = mesh.nCells
    rng = np.random.default_rng(1)
    U = 0.1 * rng.standard_normal((n, 3))
    if ptypes:
        for d in range(3):
            if ptypes[2 * d] == E:
                U[:, d] = 0.0
    T = 1.0 + 0.05 * rng.standard_normal(n)
    p = 1.0 + 0.05 * rng.standard_normal(n)
    opt = q.default_options(stencil=stencil, deltaT=1e-3, mu=1e-3)
    dev = q.Device(mesh)
    gc = q.QGDFoamCase(dev, opt)
    oc = OracleCase(oracle_mesh_of(mesh), opt)
    for c in (gc, oc):
        c.set_bc(0, U=("fixedValue", (0.1, 0.0, 0.0)), T=("fixedValue", 1.02), p=("zeroGradient", None))
        c.set_bc(1, U=("zeroGradient", None), T=("zeroGradient", None), p=("fixedValue", 1.0))
        c.set_fields(U, T, p)
    gc.updateFluxes(); oc.updateFluxes()
    for f in ("phiJm", "phiJmU", "phiPi", "phiQ", "gradUf", "gradPf"):
        assert rel_err(gc.field(f), oc.field(f)) <= 1e-11, (dims, stencil, f)
    gc.step(10); oc.step(10)
    for f in ("rho", "U", "p", "e"):
        assert rel_err(gc.field(f), oc.field(f)) <= 1e-11, (dims, stencil, f)
    assert np.all(np.isfinite(gc.field("rho")))
    gc.close(); dev.close()
