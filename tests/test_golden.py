"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them; GPU: the HIP path reproduces them through the C-ABI."""
import os
import sys

import numpy as np
import pytest

import qgdsolver_amd as q

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

from util import rel_err  # noqa: E402


@pytest.mark.parametrize("name", sorted(mg.GOLDEN))
def test_oracle_reproduces_golden(name):
    gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.run_case(mg.oracle_factory, mg.GOLDEN[name])
    assert sorted(got) == sorted(gold.files)
    for k in gold.files:
        # same code, same compiler flags (-ffp-contract=off): bit-identical on any x86-64 host
        assert np.array_equal(got[k], gold[k]), (name, k, rel_err(got[k], gold[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("arm", ["fused", "kernels"])
@pytest.mark.parametrize("name", sorted(mg.GOLDEN))
def test_gpu_reproduces_golden(name, arm):
    """both ways a device steps a case against the committed vectors: the fused one-launch step (the kernel bench.py times; 3-D GaussVolPoint
    cases) and the separate kernels -- the one that ran is asserted, not left to a default"""
    from util import assert_path, expects_fused

    gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
    devs = []

    def gpu_factory(mesh, options):
        if arm == "fused" and not expects_fused(mesh, options):
            pytest.skip("not a case the fused step serves (2-D mesh or another stencil): covered by the kernels arm")
        dev = q.Device(mesh, fused_tables="any" if arm == "fused" else False)
        devs.append(dev)
        case = q.QGDFoamCase(dev, options)
        assert_path(case, arm, name)
        return case

    got = mg.run_case(gpu_factory, mg.GOLDEN[name])
    for k in gold.files:
        assert rel_err(got[k], gold[k]) <= 1e-10, (name, k, rel_err(got[k], gold[k]))


# ---- stateless operators (fvsc grad/div, QHD flux parts, species flux block, QHD pressure equation) -------------------
def _operator_case(name):
    from util import make_mesh

    gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
    kind, scheme = mg.OPERATOR_GOLDEN[name]
    mesh = make_mesh(kind)
    inputs = {k: gold[k] for k in gold.files if k.startswith("in_")}
    want = {k[4:]: gold[k] for k in gold.files if k.startswith("out_")}
    return mesh, scheme, inputs, want


@pytest.mark.parametrize("name", sorted(mg.OPERATOR_GOLDEN))
def test_oracle_reproduces_operator_golden(name):
    mesh, scheme, inputs, want = _operator_case(name)
    # the stored inputs are what the seeded generator still produces (guards the fixture against silent regeneration)
    regenerated = mg.operator_inputs(mesh, sum(map(ord, name)))
    assert all(np.array_equal(regenerated[k], inputs[k]) for k in inputs)
    got = mg.run_operators(mg.OracleOps(mesh), mesh, scheme, inputs)
    assert sorted(got) == sorted(want)
    for k in want:
        assert np.array_equal(np.asarray(got[k]).reshape(want[k].shape), want[k]), (name, k)


class DeviceOps:
    def __init__(self, mesh):
        self.mesh = mesh
        self.dev = q.Device(mesh)

    def fvsc(self, scheme, op, cell, bnd):
        from qgdsolver_amd import fvsc
        self.dev.fvSchemes = {"fvsc": {"default": scheme}}
        vf = q.volField("f", cell, bnd)
        return fvsc.grad(self.dev, vf) if op.startswith("grad") else fvsc.div(self.dev, vf)

    def qhd(self, scheme, U, T, rho, tau, beta, g, p, phi):
        from qgdsolver_amd import qhdfoam
        return qhdfoam.updateFluxes(self.dev, scheme, U, T, rho, tau, beta, g, p=p, phi=phi)

    def species(self, scheme, Y, U, phiJm, phi, tau):
        from qgdsolver_amd import qgdfoam
        return qgdfoam.speciesFlux(self.dev, scheme, Y, U, phiJm, phi, tau)

    def pressure(self, phiu, phiwo, tbr, p0, kinds, pb, gb):
        from qgdsolver_amd import qhdfoam
        return qhdfoam.pEqn(self.dev, phiu, phiwo, tbr, p0, kinds, pb, gb, tolerance=1e-13, maxIter=5000, pRefCell=0, pRefValue=0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.OPERATOR_GOLDEN))
def test_gpu_reproduces_operator_golden(name):
    mesh, scheme, inputs, want = _operator_case(name)
    ops = DeviceOps(mesh)
    got = mg.run_operators(ops, mesh, scheme, inputs)
    for k in want:
        tol = 1e-8 if k.startswith("pEqn") else 1e-10  # the pressure equation is an iterative solve (tolerance 1e-13)
        assert rel_err(np.asarray(got[k]).reshape(want[k].shape), want[k]) <= tol, (name, k)
    ops.dev.close()
