"""BASELINE config 4 as configured: the 400^3 = 64 M-cell hex box cut 8-way into k-slabs (one ghost plane per cut), every
shard resident on the one GPU of the test box (~6 GB each), halo records moved device-to-device through the library's
pack/unpack kernels -- the data path of the 8-GPU run minus the RCCL transport itself.  Three explicit steps in the plain
order (assemble, advance, exchange) and in the overlapped order (assemble, boundary layer, pack | rest of the cells |
unpack) must reproduce the unsharded 64 M-cell run on the owned cells of every shard to <= 1e-12 (they differ by
rounding only: a cut face changes the summation order of its ghost cell)."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd.halo import slab_range

import cases

pytestmark = pytest.mark.gpu

FIELDS = ("rho", "U", "p", "e")


@pytest.mark.parametrize("n", [48, 400])
def test_eight_slab_shards_match_the_unsharded_run(n):
    world, steps = 8, 3
    opt = dict(stencil="GaussVolPoint", deltaT=0.1 / n / 1.3, mu=1e-4)
    plane = n * n

    gmesh = q.PolyMesh.box(n, n, n)
    U, T, p = cases.box_initial_fields(gmesh.array("C").reshape(-1, 3))
    gdev = q.Device(gmesh)
    gcase = q.QGDFoamCase(gdev, q.default_options(**opt))
    gcase.set_fields(U, T, p)
    gcase.step(steps)
    ref = {f: gcase.field(f) for f in FIELDS}
    assert gcase.info()["minRho"] > 0
    gcase.close(); gdev.close(); gmesh.close()
    # the state moved: the comparison below is not about an unchanged field
    assert np.abs(ref["p"] - p).max() > 1e-6

    shards = []
    for rank in range(world):
        lo, hi, k_lo, k_hi = slab_range(n, rank, world)
        mesh = q.PolyMesh.box(n, n, n, k_range=(k_lo, k_hi))
        dev = q.Device(mesh)
        case = q.QGDFoamCase(dev, q.default_options(**opt))
        shards.append(dict(rank=rank, lo=lo, hi=hi, k_lo=k_lo, k_hi=k_hi, dev=dev, case=case, sl=slice(plane * k_lo, plane * k_hi)))
        mesh.close()
    # one device buffer per direction per cut: up[r] carries shard r -> r+1, down[r] carries shard r+1 -> r
    up, down = [], []
    for r in range(world - 1):
        a, b = shards[r], shards[r + 1]
        assert a["case"].halo_count(1) == b["case"].halo_recv_count(0) >= 8 * plane
        assert b["case"].halo_count(0) == a["case"].halo_recv_count(1) >= 8 * plane
        up.append(a["dev"].alloc(8 * a["case"].halo_count(1)))
        down.append(b["dev"].alloc(8 * b["case"].halo_count(0)))
    assert shards[0]["case"].halo_count(0) == 0 and shards[-1]["case"].halo_count(1) == 0

    def pack():
        for r in range(world - 1):
            shards[r]["case"].halo_pack(1, up[r])
            shards[r + 1]["case"].halo_pack(0, down[r])

    def unpack():
        for r in range(world - 1):
            shards[r + 1]["case"].halo_unpack(0, up[r])
            shards[r]["case"].halo_unpack(1, down[r])

    def sync():
        for s in shards:
            s["case"].sync()

    for overlapped in (False, True):
        for s in shards:
            s["case"].set_fields(U[s["sl"]], T[s["sl"]], p[s["sl"]])
        pack(); sync(); unpack(); sync()
        for _ in range(steps):
            for s in shards:
                s["case"].step_phase(0)
            if not overlapped:
                for s in shards:
                    s["case"].step_phase(1)
                sync(); pack(); sync(); unpack(); sync()
            else:
                for s in shards:
                    s["case"].step_phase(10)
                sync(); pack()
                for s in shards:
                    s["case"].step_phase(11)
                sync(); unpack(); sync()
        for s in shards:
            own = slice(plane * (s["lo"] - s["k_lo"]), plane * (s["hi"] - s["k_lo"]))
            glob = slice(plane * s["lo"], plane * s["hi"])
            for f in FIELDS:
                got = s["case"].field(f)[own]
                err = np.abs(got - ref[f][glob]).max() / np.abs(ref[f]).max()
                assert err <= 1e-12, (n, "overlapped" if overlapped else "plain", s["rank"], f, err)
            assert s["case"].info()["minRho"] > 0
    for r in range(world - 1):
        shards[r]["dev"].release(up[r])
        shards[r + 1]["dev"].release(down[r])
    for s in shards:
        s["case"].close(); s["dev"].close()
