"""Cell-range (k-slab) sharding of the box and the per-step halo exchange.

One process per GPU; rank r owns the planes [lo, hi) and carries one ghost plane per cut.  Per step each
rank sends its boundary-layer records (one plane of cells + their patch faces) to each neighbour and
receives the neighbour's into its ghost plane: ONE message per neighbour per step, point-to-point (a slab has
at most two neighbours, so each pair talks over its own xGMI link).  ``torch.distributed`` is only the
transport (backend "nccl" == RCCL on ROCm, "gloo" on CPU); pack/unpack are the library's own kernels.
"""

import ctypes as C

from . import _lib as L


def slab_range(n, rank, world):
    """owned planes [lo, hi) and the mesh window [k_lo, k_hi) including ghost planes"""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    k_lo = lo - 1 if rank > 0 else lo
    k_hi = hi + 1 if rank < world - 1 else hi
    return lo, hi, k_lo, k_hi


class SlabHalo:
    """Halo exchange for a case whose mesh came from ``PolyMesh.box(..., k_range=...)``.

    ``alloc(count)`` returns a float64 torch tensor living where the case's data lives;
    ``arg(tensor)`` turns it into what ``case.halo_pack/unpack`` take (a device pointer for the HIP case).
    """

    def __init__(self, case, rank, world, dist, alloc, arg, peers=None):
        self.case, self.rank, self.world, self.dist = case, rank, world, dist
        self.arg = arg
        if peers is None:  # box slab: slot 0 = lower, slot 1 = upper neighbour
            peers = {0: rank - 1, 1: rank + 1}
        self.peer = dict(peers)
        self.sides = [s for s, peer in sorted(self.peer.items()) if 0 <= peer < world]
        self._alloc = alloc
        self.send = {s: alloc(case.halo_count(s)) for s in self.sides}
        self.recv = {s: alloc(case.halo_recv_count(s)) for s in self.sides}

    def _buffers(self, mid):
        """(send, recv, pack, unpack) of the state message or of the mid-assembly message (2 doubles per patch face of the boundary layer)"""
        if not mid:
            return self.send, self.recv, self.case.halo_pack, self.case.halo_unpack
        if getattr(self, "_mid", None) is None:
            counts = {s: self.case.mid_halo_count(s) for s in self.sides}
            self._mid = ({s: self._alloc(max(counts[s][0], 1))[:counts[s][0]] for s in self.sides},
                         {s: self._alloc(max(counts[s][1], 1))[:counts[s][1]] for s in self.sides})
        return self._mid[0], self._mid[1], self.case.mid_halo_pack, self.case.mid_halo_unpack

    def exchange(self, mid=False):
        if not self.sides:
            return
        send, recv, pack, unpack = self._buffers(mid)
        for s in self.sides:
            if len(send[s]):
                pack(s, self.arg(send[s]))
        ops = []
        for s in self.sides:
            if len(send[s]):
                ops.append(self.dist.P2POp(self.dist.isend, send[s], self.peer[s]))
            if len(recv[s]):
                ops.append(self.dist.P2POp(self.dist.irecv, recv[s], self.peer[s]))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for s in self.sides:
            if len(recv[s]):
                unpack(s, self.arg(recv[s]))

    def assemble(self):
        """step phase 0 -- in two halves with the mid-assembly message between them when the case asks for it (a GaussVolPoint
        shard that meets a qgdFlux wall: include/qgd_amd.h)"""
        if self.case.needs_mid_exchange():
            self.case.step_phase(5)
            self.exchange(mid=True)
            self.case.step_phase(6)
        else:
            self.case.step_phase(0)

    def step(self, allreduce_max=None):
        """One sharded step: assemble, (adjustTimeStep: MAX all-reduce of the 2-double reduction buffer through
        ``allreduce_max(case)``), advance, halo exchange, post-exchange refresh."""
        self.assemble()
        if allreduce_max is not None:
            allreduce_max(self.case)
        self.case.step_phase(1)
        self.exchange()
        self.case.step_phase(2)

    def step_overlapped(self, torch, compute_stream, halo_stream):
        """One sharded step with the exchange hidden behind the bulk of the cell update (fixed deltaT):
        compute stream: assemble, boundary-layer update | remaining cells ............ | next step waits
        halo stream   :                                 | pack, send/recv, unpack ...  |
        The case's kernels run on ``compute_stream`` (``case.set_stream``), pack/unpack on ``halo_stream``
        (``case.set_halo_stream``); torch.distributed orders the RCCL transfer after the halo stream's pack."""
        if getattr(self, "_halo_stream", None) is not halo_stream:
            # pack/unpack must be enqueued on the stream the RCCL transfer is ordered against
            self.case.set_halo_stream(halo_stream.cuda_stream)
            self._halo_stream = halo_stream
        self.assemble()
        self.case.step_phase(10)
        halo_stream.wait_stream(compute_stream)
        with torch.cuda.stream(halo_stream):
            self.exchange()
        self.case.step_phase(11)
        compute_stream.wait_stream(halo_stream)


class _DeviceBuffer:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_tensor(torch, ptr, n):
    """float64 torch view of ``n`` doubles of library-owned device memory (no copy), e.g. ``case.reduction_ptr()``"""
    return torch.as_tensor(_DeviceBuffer(ptr, n), device="cuda")


def allreduce_max_of(torch, dist, case, staged=False):
    """The MAX all-reduce of the {max Co, -min tauQGDf} buffer that QGDCourantNo.H / setDeltaT-QGDQHD.H need across
    shards (their ``reduce(..., maxOp)`` / ``gMin``), applied to the case's device buffer in place.  Returns the
    callable ``SlabHalo.step`` takes."""
    view = device_tensor(torch, case.reduction_ptr(), 2)

    def run(_case):
        if staged:
            _case.sync()
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX)
            view.copy_(host)
            torch.cuda.synchronize()
        else:
            dist.all_reduce(view, op=dist.ReduceOp.MAX)

    return run


class HostStaged:
    """Mixin for transports that cannot take device buffers (the gloo debugging mode: every rank may even share one
    GPU): pack into a device buffer, stage through host tensors, unpack from a device buffer.  ``alloc`` must return
    host tensors; ``torch`` is passed in by the caller."""

    torch = None

    def exchange(self, mid=False):
        if not self.sides:
            return
        torch = self.torch
        send, recv, pack, unpack = self._buffers(mid)
        dev_s = {s: torch.empty(max(send[s].numel(), 1), dtype=torch.float64, device="cuda") for s in self.sides}
        dev_r = {s: torch.empty(max(recv[s].numel(), 1), dtype=torch.float64, device="cuda") for s in self.sides}
        for s in self.sides:
            if send[s].numel():
                pack(s, dev_s[s].data_ptr())
        torch.cuda.synchronize()
        for s in self.sides:
            if send[s].numel():
                send[s].copy_(dev_s[s][:send[s].numel()])
        ops = []
        for s in self.sides:
            if send[s].numel():
                ops.append(self.dist.P2POp(self.dist.isend, send[s], self.peer[s]))
            if recv[s].numel():
                ops.append(self.dist.P2POp(self.dist.irecv, recv[s], self.peer[s]))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for s in self.sides:
            if recv[s].numel():
                dev_r[s][:recv[s].numel()].copy_(recv[s])
                unpack(s, dev_r[s].data_ptr())
        torch.cuda.synchronize()


class RangeHalo(SlabHalo):
    """Halo exchange for a case on a ``PolyMesh.shard`` mesh (any polyMesh cut into contiguous cell ranges): one
    message per neighbouring rank per step, the neighbours being whoever shares a vertex with the owned range."""

    def __init__(self, case, rank, world, dist, alloc, arg):
        peers = {slot: int(p) for slot, p in enumerate(case.mesh.array("haloPeer"))}
        super().__init__(case, rank, world, dist, alloc, arg, peers=peers)


class NativeComm:
    """qgd_comm_t: the library's own RCCL communicator (qgd_comm_* / qgd_case_halo_exchange / qgd_case_step_sharded in
    include/qgd_amd.h) -- what a C++/MPI host uses; this wrapper bootstraps it the way such a host would, with the launcher's
    broadcast standing in for MPI_Bcast (``bcast(bytes_or_None) -> bytes`` on every rank; identity on one rank)."""

    def __init__(self, device_id, rank=0, world=1, bcast=None):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(L.lib.qgd_comm_unique_id(uid), "qgd_comm_unique_id")
        raw = bytes(uid.raw)
        if world > 1:
            raw = bcast(raw if rank == 0 else None)
        buf = (C.c_char * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        L.check(L.lib.qgd_comm_create(int(device_id), int(rank), int(world), buf, C.byref(h)), "qgd_comm_create")
        self._h, self.rank, self.world = h, rank, world

    def info(self):
        """{rank, ranks, device} as RCCL itself reports them for the communicator (qgd_comm_info)"""
        v = (self._C.c_int32 * 3)()
        self._L.check(self._L.lib.qgd_comm_info(self._h, v), "qgd_comm_info")
        return {"rank": int(v[0]), "ranks": int(v[1]), "device": int(v[2])}

    def _peers(self, peers):
        import numpy as np
        a = np.ascontiguousarray(peers, dtype=np.int32)
        return a, a.ctypes.data_as(self._L.c_int32_p), int(a.size)

    def exchange(self, case, peers):
        a, p, n = self._peers(peers)
        self._L.check(self._L.lib.qgd_case_halo_exchange(case._h, self._h, p, n), "qgd_case_halo_exchange")

    def step(self, case, peers, overlapped=False):
        a, p, n = self._peers(peers)
        self._L.check(self._L.lib.qgd_case_step_sharded(case._h, self._h, p, n, 1 if overlapped else 0), "qgd_case_step_sharded")

    def allreduce_max(self, case):
        self._L.check(self._L.lib.qgd_case_allreduce_max(case._h, self._h), "qgd_case_allreduce_max")

    def qhd_exchange(self, case, peers, kind=0):
        """message kind 0 (state), 1 (p) ... of a sharded QHDFoam case over RCCL (qgd_qhd_case_halo_exchange)"""
        a, p, n = self._peers(peers)
        self._L.check(self._L.lib.qgd_qhd_case_halo_exchange(case._h, self._h, p, n, int(kind)), "qgd_qhd_case_halo_exchange")

    def qhd_step(self, case, peers, n_steps=1):
        """whole QHDFoam steps of a sharded case with the transport inside the library (qgd_qhd_case_step_sharded): halo messages,
        the all-reduced scalars of the pressure solve and the comm points of its multigrid hierarchy, all on the device's stream"""
        a, p, n = self._peers(peers)
        self._L.check(self._L.lib.qgd_qhd_case_step_sharded(case._h, self._h, p, n, int(n_steps)), "qgd_qhd_case_step_sharded")

    def close(self):
        if getattr(self, "_h", None):
            self._L.lib.qgd_comm_free(self._h)
            self._h = None


# ---- QHDFoam on cell-range shards ----------------------------------------------------------------------------------------
# control-block slots that are global sums (include/qgd_amd.h "the QHD case on a cell-range shard")
QHD_REDUCE_AFTER_PHASE = {0: (0, 3), 1: (3, 1), 2: (4, 1), 3: (5, 1), 4: (6, 2), 7: (8, 1)}
QHD_STATE, QHD_PRESSURE, QHD_DIRECTION, QHD_MG_ITERATE, QHD_IMPLICIT_ITERATE = 0, 1, 2, 3, 4   # halo message kinds


class QhdStepper:
    """The choreography of one QHDFoam step on cell-range shards: the phases of ``qgd_qhd_case_step_phase`` with, between
    them, the SUM reductions of the control block and the halo messages.  ``world`` supplies the three verbs:

    ``phase(k)`` runs phase k on every shard this process holds, ``allreduce(first, count)`` sums control[first:first+count]
    over ALL shards, ``exchange(kind)`` moves message kind (0 state, 1 pressure, 2 search direction) between neighbouring
    shards, ``done()`` tells whether the pressure solve has finished (the same answer on every shard: it is computed from
    reduced sums).  ``LocalWorld``: several shards in one process; ``DistWorld``: one shard per rank over torch.distributed.
    """

    def __init__(self, world):
        self.world = world
        self.started = False

    def start(self):
        """after set_fields: the ghost cells' state from their owners (set_fields gave them values already; this makes the
        start independent of what the caller put there)"""
        self.world.exchange(QHD_STATE)
        self.started = True

    def step(self, n=1):
        w = self.world
        if not self.started:
            self.start()
        for _ in range(n):
            for k in (0, 1, 2):
                w.phase(k)
                w.allreduce(*QHD_REDUCE_AFTER_PHASE[k])
            w.exchange(QHD_DIRECTION)
            while not w.done():
                for k in (3, 4):
                    w.phase(k)
                    w.allreduce(*QHD_REDUCE_AFTER_PHASE[k])
                w.phase(5)
                w.exchange(QHD_DIRECTION)
            w.phase(6)
            w.exchange(QHD_PRESSURE)
            w.phase(7)
            if w.implicit():
                # implicitDiffusion: the four systems {Ux, Uy, Uz, T} as one solve with its own control block; phases 10..15 are the
                # solver phases of the QGDFoam branch (22..27 there), message kind 4 carries what the next product reads in the ghosts
                for sp in (0, 1, 2):
                    w.phase(10 + sp)
                    w.allreduce(*IMPL_REDUCE_AFTER_SOLVER_PHASE[sp], block="implicit")
                    if sp == 0:
                        w.allreduce(*IMPL_MAX_AFTER_SOLVER_PHASE_0, op="max", block="implicit")
                w.exchange(QHD_IMPLICIT_ITERATE)
                while not w.done(block="implicit"):
                    for sp in (3, 4):
                        w.phase(10 + sp)
                        w.allreduce(*IMPL_REDUCE_AFTER_SOLVER_PHASE[sp], block="implicit")
                    w.phase(15)
                    w.exchange(QHD_IMPLICIT_ITERATE)
                w.phase(16)
            w.allreduce(*QHD_REDUCE_AFTER_PHASE[7])
            w.phase(8)
            w.exchange(QHD_STATE)


class LocalWorld:
    """All shards in this process (``cases[r]`` is the case of rank r, ``peers[r][slot]`` the rank behind each of its halo
    slots, < 0 for none): messages go pack -> buffer -> unpack, reductions through host copies of the control blocks."""

    def __init__(self, cases, peers, kinds=(0, 1, 2)):
        self.cases, self.peers = list(cases), [list(p) for p in peers]
        if self.needs_mid():
            kinds = tuple(kinds) + (IMPL_MID,)
        self.buf = {}
        for r, case in enumerate(self.cases):
            for slot, peer in enumerate(self.peers[r]):
                if peer < 0:
                    continue
                n = max(case.halo_count(slot, kind)[0] for kind in kinds)
                self.buf[(r, slot)] = case.halo_buffer(n)

    def needs_mid(self):
        return any(getattr(c, "needs_mid_exchange", lambda: False)() for c in self.cases)

    def phase(self, k):
        for c in self.cases:
            c.step_phase(k)
        self._drain()

    def _drain(self):
        """the comm points inside a phase (the multigrid hierarchy that spans the ranks: qgd_qhd_case_pending)"""
        if not hasattr(self.cases[0], "pending"):
            return
        import numpy as np
        while True:
            pend = [c.pending() for c in self.cases]
            assert len({p[0] for p in pend}) == 1 and len({p[2] for p in pend}) == 1, pend
            action, _, count = pend[0]
            if action == 0:
                return
            if action == 1:
                self.exchange(QHD_MG_ITERATE)
            else:
                for c in self.cases:
                    c.sync()                      # the buffers are written by kernels on the cases' streams
                host = [c.dev.to_host(p[1], (count,)) for c, p in zip(self.cases, pend)]
                total = np.sum(host, axis=0) if action == 2 else np.max(host, axis=0)
                for c, p in zip(self.cases, pend):
                    L.check(L.lib.qgd_device_copy(c.dev._h, C.c_void_p(p[1]), total.ctypes.data_as(C.c_void_p), total.nbytes, 1), "qgd_device_copy")
            for c in self.cases:
                c.step_phase(9)

    def implicit(self):
        return bool(getattr(self.cases[0], "implicit", False))

    def allreduce(self, first, count, op="sum", block=None):
        """block="implicit": the control block of a QHD case's implicitDiffusion solve instead of the pressure solve's"""
        import numpy as np
        get = (lambda c: c.implicit_control()) if block == "implicit" else (lambda c: c.control())
        ctl = [get(c) for c in self.cases]
        total = sum(a[first:first + count] for a in ctl) if op == "sum" else np.max([a[first:first + count] for a in ctl], axis=0)
        for c, a in zip(self.cases, ctl):
            a[first:first + count] = total
            c.set_implicit_control(a) if block == "implicit" else c.set_control(a)

    def exchange(self, kind):
        for (r, slot), buf in self.buf.items():
            self.cases[r].halo_pack(slot, kind, buf)
        for c in self.cases:
            c.sync()
        for (r, slot), buf in self.buf.items():
            peer = self.peers[r][slot]
            back = self.peers[peer].index(r)           # the peer's slot towards r
            assert self.cases[r].halo_count(slot, kind)[0] == self.cases[peer].halo_count(back, kind)[1]
            self.cases[peer].halo_unpack(back, kind, buf)
        for c in self.cases:
            c.sync()

    def done(self, block=None):
        flags = [(1 if c.implicit_solve_done() else 0) if block == "implicit" else c.solve_status()["done"] for c in self.cases]
        assert all(f == flags[0] for f in flags), flags
        return flags[0] != 0


class DistWorld:
    """One shard per rank over ``torch.distributed`` (backend "nccl" == RCCL, or "gloo" with host-staged buffers):
    ``peers[slot]`` = rank behind each halo slot.  ``to_transport(buf, n)`` / ``from_transport(t, buf)`` turn the case's
    halo buffer (a device pointer, or a host array when the cases live on the CPU) into a torch tensor the backend can send and back."""

    def __init__(self, case, dist, torch, peers, to_transport, from_transport, kinds=(0, 1, 2), device_reduce=False):
        # device_reduce: the backend takes device tensors (nccl == RCCL) -- reductions run in place on views of the library's own
        # device memory (the case's kernels must run on torch's current stream: case.set_stream)
        self.case, self.dist, self.torch, self.peers = case, dist, torch, list(peers)
        self.device_reduce = device_reduce
        self.to_transport, self.from_transport = to_transport, from_transport
        self.slots = [s for s, p in enumerate(self.peers) if p >= 0]
        if self.needs_mid():
            kinds = tuple(kinds) + (IMPL_MID,)
        self.sbuf = {s: case.halo_buffer(max(case.halo_count(s, k)[0] for k in kinds)) for s in self.slots}
        self.rbuf = {s: case.halo_buffer(max(case.halo_count(s, k)[1] for k in kinds)) for s in self.slots}

    def needs_mid(self):
        return getattr(self.case, "needs_mid_exchange", lambda: False)()

    def phase(self, k):
        self.case.step_phase(k)
        if not hasattr(self.case, "pending"):
            return
        while True:                       # the comm points inside a phase (qgd_qhd_case_pending)
            action, ptr, count = self.case.pending()
            if action == 0:
                return
            if action == 1:
                self.exchange(QHD_MG_ITERATE)
            elif self.device_reduce:
                t = device_tensor(self.torch, ptr, count)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if action == 2 else self.dist.ReduceOp.MAX)
            else:
                self.case.sync()                  # the buffer is written by kernels on the case's stream
                t = self.to_transport(ptr, count)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if action == 2 else self.dist.ReduceOp.MAX)
                self.from_transport(t, ptr)
                self.case.sync()
            self.case.step_phase(9)

    def implicit(self):
        return bool(getattr(self.case, "implicit", False))

    def allreduce(self, first, count, op="sum", block=None):
        rop = self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX
        imp = block == "implicit"
        if self.device_reduce:
            t = device_tensor(self.torch, (self.case.implicit_control_ptr() if imp else self.case.control_ptr()) + 8 * first, count)
            self.dist.all_reduce(t, op=rop)
            return
        a = self.case.implicit_control() if imp else self.case.control()
        t = self.torch.from_numpy(a[first:first + count].copy())
        self.dist.all_reduce(t, op=rop)
        a[first:first + count] = t.numpy()
        self.case.set_implicit_control(a) if imp else self.case.set_control(a)

    def exchange(self, kind):
        if not self.slots:
            return
        ops, recv = [], {}
        for s in self.slots:
            ns, nr = self.case.halo_count(s, kind)
            self.case.halo_pack(s, kind, self.sbuf[s])
        self.case.sync()
        for s in self.slots:
            ns, nr = self.case.halo_count(s, kind)
            if ns:
                ops.append(self.dist.P2POp(self.dist.isend, self.to_transport(self.sbuf[s], ns), self.peers[s]))
            if nr:
                recv[s] = self.to_transport(self.rbuf[s], nr)
                ops.append(self.dist.P2POp(self.dist.irecv, recv[s], self.peers[s]))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for s, t in recv.items():
            self.from_transport(t, self.rbuf[s])
            self.case.halo_unpack(s, kind, self.rbuf[s])
        self.case.sync()

    def done(self, block=None):
        if block == "implicit":
            return self.case.implicit_solve_done()
        return self.case.solve_status()["done"] != 0


# ---- QGDFoam with implicitDiffusion true (the reference's default) on cell-range shards ---------------------------------------
# control-block ranges (slot-major, 4 components per slot) that are global sums, by SOLVER phase 0..4
IMPL_REDUCE_AFTER_SOLVER_PHASE = {0: (0, 12), 1: (12, 4), 2: (16, 4), 3: (20, 4), 4: (24, 8)}
IMPL_MAX_AFTER_SOLVER_PHASE_0 = (32, 4)   # control slot 8: max over the ranks (Gershgorin radius of the Chebyshev solves)
IMPL_STATE, IMPL_GRADU, IMPL_U, IMPL_DIRECTION, IMPL_GUESS = 0, 1, 2, 3, 4   # message kinds (0: the case's own state message)
IMPL_MID = 5   # the message in the middle of the flux assembly (qgd_case_mid_halo_*)


class ImplicitShard:
    """A QGDFoam case (device or CPU) with implicitDiffusion true, seen through the names LocalWorld / DistWorld use: message
    kind 0 is the case's state message (qgd_case_halo_*), kinds 1..4 the branch's own (qgd_case_implicit_halo_*); the control
    block is the one of the linear solve in flight."""

    def __init__(self, case):
        self.case = case

    def step_phase(self, k):
        self.case.step_phase(k)

    def control(self):
        return self.case.implicit_control()

    def set_control(self, a):
        self.case.set_implicit_control(a)

    def control_ptr(self):
        return self.case.implicit_control_ptr()

    def solve_status(self):
        return dict(done=1 if self.case.implicit_solve_done() else 0)

    def sync(self):
        self.case.sync()

    def halo_buffer(self, n):
        return self.case.halo_buffer(n)

    def needs_mid_exchange(self):
        return self.case.needs_mid_exchange()

    def halo_count(self, slot, kind):
        if kind == IMPL_STATE:
            return self.case.halo_count(slot), self.case.halo_recv_count(slot)
        if kind == IMPL_MID:
            return self.case.mid_halo_count(slot)
        return self.case.implicit_halo_count(slot, kind)

    def halo_pack(self, slot, kind, buf):
        if kind == IMPL_STATE:
            self.case.halo_pack(slot, buf)
        elif kind == IMPL_MID:
            self.case.mid_halo_pack(slot, buf)
        else:
            self.case.implicit_halo_pack(slot, kind, buf)

    def halo_unpack(self, slot, kind, buf):
        if kind == IMPL_STATE:
            self.case.halo_unpack(slot, buf)
        elif kind == IMPL_MID:
            self.case.mid_halo_unpack(slot, buf)
        else:
            self.case.implicit_halo_unpack(slot, kind, buf)


class ImplicitStepper:
    """One QGDFoam step with implicitDiffusion true on cell-range shards: phase 0 (flux assembly), then the advance as phases
    20..35 with the reductions and messages include/qgd_amd.h lists.  ``world``: LocalWorld / DistWorld built over
    ``ImplicitShard`` wrappers (their message kinds must cover 0..4: pass ``kinds=range(5)``)."""

    def __init__(self, world):
        self.world = world
        self.started = False

    def _solve(self):
        w = self.world
        w.exchange(IMPL_GUESS)
        for sp in (0, 1, 2):
            w.phase(22 + sp)
            w.allreduce(*IMPL_REDUCE_AFTER_SOLVER_PHASE[sp])
            if sp == 0:
                w.allreduce(*IMPL_MAX_AFTER_SOLVER_PHASE_0, op="max")   # the Chebyshev solves' spectral bound (zeros under PCG)
        w.exchange(IMPL_DIRECTION)
        while not w.done():
            for sp in (3, 4):
                w.phase(22 + sp)
                w.allreduce(*IMPL_REDUCE_AFTER_SOLVER_PHASE[sp])
            w.phase(27)
            w.exchange(IMPL_DIRECTION)

    def step(self, n=1, allreduce_max=None):
        """``allreduce_max()`` (adjustTimeStep): the MAX all-reduce of every shard's {max Co, -min tauQGDf} buffer, between the
        assembly and phase 20 (which forms deltaT from it)"""
        w = self.world
        if not self.started:
            w.exchange(IMPL_STATE)     # ghost cells start from their owners' records
            w.phase(2)
            self.started = True
        for _ in range(n):
            if w.needs_mid():          # GaussVolPoint shards meeting a qgdFlux wall: the assembly in two halves
                w.phase(5)
                w.exchange(IMPL_MID)
                w.phase(6)
            else:
                w.phase(0)
            if allreduce_max is not None:
                allreduce_max()
            w.phase(20)
            w.exchange(IMPL_GRADU)
            w.phase(21)
            self._solve()
            w.phase(28)
            w.exchange(IMPL_U)
            w.phase(29)
            w.exchange(IMPL_GRADU)
            w.phase(30)
            self._solve()
            w.phase(35)
            w.exchange(IMPL_STATE)
            w.phase(2)
