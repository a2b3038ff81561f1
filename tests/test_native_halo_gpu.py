"""The library's own RCCL halo transport (qgd_comm_*, qgd_case_halo_exchange, qgd_case_step_sharded) on the one GPU of the
test box: a one-rank communicator, a no-op on an unsharded case, and the pack -> ncclSend/ncclRecv to self -> unpack loop on
the middle slab of a three-slab cut (its two halo slots talk to rank 0 = itself, so the ghost planes receive the case's own
boundary layers: not a physical set-up, but every byte goes through the packed message, RCCL and the unpack kernel).
N > 1 ranks over xGMI cannot run here (one GPU per box): that leg is covered by the driver's scaling run only."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd.halo import NativeComm

import cases

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_and_unsharded_noop():
    comm = NativeComm(0)
    mesh = q.PolyMesh.box(8, 7, 6)
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    ref = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3))
    ref.set_fields(U, T, p)
    ref.step(4)
    for _ in range(4):
        comm.step(case, [], overlapped=False)     # unsharded: assemble + advance, nothing to exchange
    case.sync()
    comm.allreduce_max(case)                      # one rank: no-op
    for f in ("rho", "U", "p"):
        assert np.array_equal(case.field(f), ref.field(f)), f
    case.close(); ref.close(); dev.close(); comm.close()


@pytest.mark.parametrize("overlapped", [False, True])
def test_self_exchange_moves_the_boundary_layers_through_rccl(overlapped):
    nx, ny, n = 9, 8, 12
    plane = nx * ny
    comm = NativeComm(0)
    mesh = q.PolyMesh.box(nx, ny, n, k_range=(3, 9))          # planes 3..8: ghost 3, owned 4..7, ghost 8
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    assert case.halo_count(0) == case.halo_recv_count(0) == case.halo_count(1) == case.halo_recv_count(1) > 8 * plane
    comm.exchange(case, [0, 0])
    case.sync()
    rho, Uf = case.field("rho").reshape(6, plane), case.field("U").reshape(6, plane, 3)
    # sends to self are matched with receives in issue order: slot 0's message (lowest owned plane) lands in slot 0's ghost plane
    assert np.array_equal(rho[0], rho[1]) and np.array_equal(Uf[0], Uf[1])
    assert np.array_equal(rho[5], rho[4]) and np.array_equal(Uf[5], Uf[4])
    # steps through the library's own choreography stay finite and keep moving the layers
    for _ in range(3):
        comm.step(case, [0, 0], overlapped=overlapped)
    case.sync()
    rho = case.field("rho").reshape(6, plane)
    assert np.isfinite(rho).all() and np.array_equal(rho[0], rho[1]) and np.array_equal(rho[5], rho[4])
    assert case.info()["minRho"] > 0
    case.close(); dev.close(); comm.close()


def test_two_slots_towards_one_other_rank_are_refused():
    """RCCL matches the messages of a peer in issue order: two halo slots with the same (other) peer rank would land in each
    other's ghost lists, so the exchange refuses them; so does a peer outside the communicator"""
    comm = NativeComm(0)
    mesh = q.PolyMesh.box(9, 8, 12, k_range=(3, 9))
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    U, T, p = cases.box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    for peers in ([1, 1], [0, 5]):
        with pytest.raises(q.QgdError) as ei:
            comm.exchange(case, peers)
        assert ei.value.code == q._lib.ERR_INVALID
    comm.exchange(case, [0, 0])     # the communicator is still usable: no group was left open
    case.sync()
    case.close(); dev.close(); comm.close()
