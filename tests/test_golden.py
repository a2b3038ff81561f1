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
@pytest.mark.parametrize("name", sorted(mg.GOLDEN))
def test_gpu_reproduces_golden(name):
    gold = np.load(os.path.join(HERE, "golden", name + ".npz"))
    devs = []

    def gpu_factory(mesh, options):
        dev = q.Device(mesh)
        devs.append(dev)
        return q.QGDFoamCase(dev, options)

    got = mg.run_case(gpu_factory, mg.GOLDEN[name])
    for k in gold.files:
        assert rel_err(got[k], gold[k]) <= 1e-10, (name, k, rel_err(got[k], gold[k]))
