"""Per-term entries of fvSchemes.fvsc in the resident QGDFoam case (VERDICT r04 missing #4; fvsc.C L51-58).

``fvsc{default GaussVolPoint; grad(p) reduced;}`` gives the four face gradients of updateFluxes.H L41-65 different stencils.
The case serves up to two distinct ones (``qgd_case_options::termStencil``): each gradient by its own stencil, the flux algebra
unchanged; GaussVolPoint's re-evaluation of its input's boundary conditions (quirk B6) follows grad(p)'s word.

CPU: the oracle's mixed case takes every gradient from the uniform case of that term's word, bit for bit.  GPU: device parity
against the oracle on 3-D and 2-D meshes incl. the qgdFlux walls of the step, plus the refusals."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L

import cases
from oracle import OracleCase
from util import make_mesh, oracle_mesh_of, rel_err

GRAD_OF = {"grad(U)": "gradUf", "grad(e)": "gradef", "grad(rho)": "gradRhof", "grad(p)": "gradPf"}


def _oracle(mesh, om, stencil, terms, bc_fn, init_fn, **kw):
    oc = OracleCase(om, q.default_options(stencil=stencil, termStencils=terms, **kw))
    if bc_fn:
        bc_fn(oc)
    oc.set_fields(*init_fn(mesh.array("C").reshape(-1, 3)))
    return oc


@pytest.mark.parametrize("kind,default,terms", [
    ("box654_jitter", "GaussVolPoint", {"grad(p)": "reduced"}),
    ("box654_jitter", "reduced", {"grad(U)": "GaussVolPoint", "grad(e)": "GaussVolPoint"}),
    ("plane2d_jitter", "leastSquares", {"grad(p)": "GaussVolPoint", "grad(rho)": "GaussVolPoint"}),
])
def test_oracle_mixed_case_takes_each_gradient_from_its_own_stencil(kind, default, terms):
    import test_case_parity_gpu as t
    mesh = make_mesh(kind)
    om = oracle_mesh_of(mesh)
    two_d = mesh.nGeometricD == 2
    bc_fn, init_fn = (t.empty_z_bcs, t.plane_init) if two_d else (None, cases.box_initial_fields)
    mixed = _oracle(mesh, om, default, terms, bc_fn, init_fn, deltaT=1e-3, mu=1e-3)
    mixed.updateFluxes()
    uniform = {}
    for word in {default, *terms.values()}:
        uniform[word] = _oracle(mesh, om, word, None, bc_fn, init_fn, deltaT=1e-3, mu=1e-3)
        uniform[word].updateFluxes()
    for term, name in GRAD_OF.items():
        word = terms.get(term, default)
        assert np.array_equal(mixed.field(name), uniform[word].field(name)), (term, word)
        other = [w for w in uniform if w != word][0]
        assert not np.array_equal(mixed.field(name), uniform[other].field(name)), (term, "is not the other stencil's")
    # the fluxes built on them belong to neither uniform case
    assert all(not np.array_equal(mixed.field("phiJm"), u.field("phiJm")) for u in uniform.values())
    # all terms naming the default's word IS the uniform case
    same = _oracle(mesh, om, default, {k: default for k in GRAD_OF}, bc_fn, init_fn, deltaT=1e-3, mu=1e-3)
    same.step(3); uniform[default].step(3)
    for name in ("rho", "U", "p", "e"):
        assert np.array_equal(same.field(name), uniform[default].field(name)), name


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
MIXED = [
    ("box654_jitter", "GaussVolPoint", {"grad(p)": "reduced"}, "mixed_box_bcs", None, dict(deltaT=5e-4, mu=2e-3)),
    ("box654_poly", "GaussVolPoint", {"grad(e)": "reduced", "grad(rho)": "reduced"}, None, None, dict(deltaT=1e-3, mu=1e-3)),
    ("box654_tri", "reduced", {"grad(U)": "GaussVolPoint"}, None, None, dict(deltaT=1e-3)),
    ("plane2d_jitter", "leastSquares", {"grad(p)": "GaussVolPoint"}, "empty_z_bcs", "plane_init", dict(deltaT=5e-4, mu=1e-3)),
    ("plane2d", "GaussVolPoint", {"grad(U)": "leastSquares"}, "empty_z_bcs", "plane_init", dict(deltaT=1e-3)),
    ("plane2d", "reduced", {"grad(p)": "leastSquares", "grad(e)": "leastSquares"}, "empty_z_bcs", "plane_init", dict(deltaT=1e-3)),
    # the step's qgdFlux walls: grad(p) by GaussVolPoint re-evaluates p's boundary conditions in the middle of the assembly (B6) ...
    ("step2d", "leastSquares", {"grad(p)": "GaussVolPoint"}, "forward_step_bcs", "step_init", dict(deltaT=5e-4)),
    # ... grad(p) by leastSquares does not, whatever the other gradients use
    ("step2d", "GaussVolPoint", {"grad(p)": "leastSquares"}, "forward_step_bcs", "step_init", dict(deltaT=5e-4)),
    ("step2d", "GaussVolPoint", {"grad(p)": "reduced"}, "forward_step_bcs", "step_init", dict(deltaT=5e-4, implicitDiffusion=1, implicitTol=1e-14, mu=1e-3)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,default,terms,bc,init,opt", MIXED)
def test_device_mixed_stencil_case_matches_the_oracle(kind, default, terms, bc, init, opt):
    import test_case_parity_gpu as t
    bc_fn = {None: None, "mixed_box_bcs": t.mixed_box_bcs, "empty_z_bcs": t.empty_z_bcs, "forward_step_bcs": cases.forward_step_bcs}[bc]
    init_fn = {None: None, "plane_init": t.plane_init, "step_init": t.step_init}[init]
    mesh, dev, gc, oc = t.build_pair(kind, default, bc_fn, init_fn, termStencils=terms, **opt)
    gc.updateFluxes(); oc.updateFluxes()
    t.compare_fields(gc, oc, t.FACE_FIELDS, t.FLUX_TOL, (kind, default, str(terms), "fluxes"))
    for chunk in (1, 9):
        gc.step(chunk); oc.step(chunk)
        t.compare_fields(gc, oc, ["rho", "U", "p", "e", "rhoU", "rhoE", "p.boundary"], t.STATE_TOL, (kind, default, str(terms), f"step+{chunk}"))
    gc.close(); dev.close()


@pytest.mark.gpu
def test_device_mixed_case_with_upwind_fluxes_and_adjustable_time_step():
    import test_case_parity_gpu as t
    mesh, dev, gc, oc = t.build_pair("box654_jitter", "GaussVolPoint", t.mixed_box_bcs, None, termStencils={"grad(p)": "reduced"}, deltaT=1e-4,
                                     mu=2e-3, fluxSchemeU=1, fluxSchemeH=1, adjustTimeStep=1, maxCo=0.3, maxDeltaT=1.0, cTau=0.75)
    gc.step(8); oc.step(8)
    t.compare_fields(gc, oc, ["rho", "U", "p", "e"], t.STATE_TOL, ("mixed + upwind + adjustTimeStep",))
    assert abs(gc.info()["deltaT"] - oc.info()["deltaT"]) <= 1e-12 * oc.info()["deltaT"]
    gc.close(); dev.close()


@pytest.mark.gpu
def test_device_refuses_what_it_cannot_serve():
    mesh = make_mesh("plane2d")
    dev = q.Device(mesh)
    with pytest.raises(L.QgdError, match="more than two distinct") as e:
        q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", termStencils={"grad(p)": "reduced", "grad(e)": "leastSquares"}))
    assert e.value.code == L.ERR_NOT_IMPLEMENTED
    dev.close()
    mesh = make_mesh("box654")
    dev = q.Device(mesh)
    with pytest.raises(L.QgdError) as e:   # fvscOpName's check applies per term: leastSquares in 3-D [fvsc.C L60-63]
        q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", termStencils={"grad(p)": "leastSquares"}))
    assert e.value.code == L.ERR_SCHEME
    # every term naming the default's word is the uniform case (the fused kernels)
    a = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, termStencils={k: "GaussVolPoint" for k in GRAD_OF}))
    b = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    fields = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    for c in (a, b):
        c.set_fields(*fields)
        c.step(4)
    for name in ("rho", "U", "p", "e"):
        assert np.array_equal(a.field(name), b.field(name)), name
    a.close(); b.close(); dev.close()
