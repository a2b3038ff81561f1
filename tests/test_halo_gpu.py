"""GPU: the sharded HIP path (two k-slab shards on one device, halo records moved device-to-device through the
library's pack/unpack kernels) reproduces the unsharded HIP run and the oracle."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd.halo import slab_range

import cases
from oracle import OracleCase
from util import oracle_mesh_of, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("split", [False, True])
def test_two_shards_on_one_device_match_unsharded(split):
    nx, ny, n, steps = 10, 9, 12, 8
    opt = q.default_options(stencil="GaussVolPoint", deltaT=2e-3, mu=1e-3)
    gmesh = q.PolyMesh.box(nx, ny, n)
    U, T, p = cases.box_initial_fields(gmesh.array("C").reshape(-1, 3))
    gdev = q.Device(gmesh)
    gcase = q.QGDFoamCase(gdev, opt)
    gcase.set_fields(U, T, p)
    gcase.step(steps)
    oc = OracleCase(oracle_mesh_of(gmesh), opt)
    oc.set_fields(U, T, p)
    oc.step(steps)

    plane = nx * ny
    world = 2
    shards = []
    for rank in range(world):
        lo, hi, k_lo, k_hi = slab_range(n, rank, world)
        mesh = q.PolyMesh.box(nx, ny, n, k_range=(k_lo, k_hi))
        dev = q.Device(mesh)
        case = q.QGDFoamCase(dev, opt)
        sl = slice(plane * k_lo, plane * k_hi)
        case.set_fields(U[sl], T[sl], p[sl])
        shards.append((lo, hi, k_lo, k_hi, mesh, dev, case))
    c0, c1 = shards[0][6], shards[1][6]
    with pytest.raises(q.QgdError):
        c0.step(1)  # a sharded case must be driven through step_phase + exchange
    assert c0.halo_count(0) == 0 and c1.halo_count(1) == 0
    assert c0.halo_count(1) == c1.halo_count(0) > 8 * plane
    b01 = shards[0][5].alloc(8 * c0.halo_count(1))
    b10 = shards[1][5].alloc(8 * c1.halo_count(0))

    def exchange():
        c0.halo_pack(1, b01); c1.halo_pack(0, b10)
        c0.sync(); c1.sync()
        c1.halo_unpack(0, b01); c0.halo_unpack(1, b10)
        c0.sync(); c1.sync()

    exchange()
    for _ in range(steps):
        c0.step_phase(0); c1.step_phase(0)
        if not split:
            c0.step_phase(1); c1.step_phase(1)
            exchange()
        else:
            # boundary layer first, pack, then the rest of the cells, then unpack: the order the overlapped exchange uses
            c0.step_phase(10); c1.step_phase(10)
            c0.halo_pack(1, b01); c1.halo_pack(0, b10)
            c0.step_phase(11); c1.step_phase(11)
            c0.sync(); c1.sync()
            c1.halo_unpack(0, b01); c0.halo_unpack(1, b10)
            c0.sync(); c1.sync()
    for lo, hi, k_lo, k_hi, mesh, dev, case in shards:
        own = slice(plane * (lo - k_lo), plane * (hi - k_lo))
        for f in ("rho", "U", "p", "e"):
            got = case.field(f)[own]
            assert rel_err(got, gcase.field(f)[plane * lo: plane * hi]) <= 1e-13, f
            assert rel_err(got, oc.field(f)[plane * lo: plane * hi]) <= 1e-10, f
        info = case.info()
        assert info["minRho"] > 0
    shards[0][5].release(b01)
    shards[1][5].release(b10)


def test_overlapped_choreography():
    """compute stream + halo stream with events, as bench.py drives the multi-GPU step (run in a subprocess so that torch
    initialises the HIP runtime first)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "overlap_worker.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    worst = float([ln for ln in r.stdout.splitlines() if ln.startswith("OVERLAP_WORST")][-1].split()[1])
    assert worst <= 1e-13
