#!/usr/bin/env python3
"""bench.py -- QGDFoam explicit step throughput (Mcell-steps/s) on N MI355X GPUs of one node.

Workload: BASELINE.json config "QGDFoam 64M-cell hex box" family: box [0,1]^3 of n^3
uniform hex cells in blockMesh numbering, all six patches zeroGradient, fvsc GaussVolPoint, constScPrModel1
(Sc=Pr=1, alphaQGD=0.5), explicit diffusion, fixed deltaT, seeded noise.  The mesh shards by cell range
(k-slabs) over the ranks: strong scaling, total work fixed; one RCCL halo exchange of ghost-cell records per
step (one cell plane + its patch faces per neighbour).

A step = one pass of the QGDFoam loop body (flux assembly + cell update + BC refresh), inputs resident in HBM.
Prints ONE JSON line on rank 0.  Beside the contract's keys the line carries
  N = 1: `secondary` {qhd_n200, qhd_implicit_n200, implicit_n200, qhd_c5} -- the QHDFoam step (both branches of implicitDiffusion), QGDFoam's
         implicitDiffusion step at 200^3 and the QHDFoam step on BASELINE config 5's 16 M-cell irregular mesh, 20 timed steps each, as
         child processes after the headline (--no-secondary skips them; skipped by themselves under a profiler), `dropin_fvsc`,
         `cpu_baseline`;
  N > 1: `native_transport` {ms_per_step, value, checksum_rho, rccl_ranks} -- the same run repeated by a second, fresh set of ranks
         over the library's own RCCL path (qgd_case_step_sharded, the C-ABI a C++/MPI host calls; --no-native-line skips it);
  always: `config.rccl_ranks` (ranks the halo communicator reports), `config.env` (every QGD_* variable that was set).
Without a launcher around it `--gpus N` starts its own ranks; that relay parent never touches the GPU and enforces
QGD_BENCH_DEADLINE_S (default 900 s: children killed, status 124).  Ranks started by torch.distributed.run carry the same
deadline as a watchdog timer.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# ALGORITHMIC bytes of the fused face kernel as SURVEY.md 8(d) defines them (the figure `roofline.achieved` is quoted on):
# per internal face 144 B streamed in (owner+neighbour 8, Sf 24, weight 8, hQGDf 8, 9 Gauss coefficients + 1/V 80, 4 vertex
# labels 16) + 40 B net fluxes out; every cell and every vertex record (rho,U,p,e = 48 B) gathered once.
FACE_BYTES_PER_FACE = 184
FACE_BYTES_PER_CELL = 48
FACE_BYTES_PER_POINT = 48
# What the kernel that is actually launched (faceFluxGvp3TileKernel) has to move at least once, in its own layout: it does not
# stream the 80 B of Gauss coefficients but rebuilds them from gathered geometry, and it addresses its records through per-tile
# lists of distinct labels.  Per internal face: six 16-bit list positions 12, flux position 4, kind 1, weight 8, hQGDf 8
# = 33 B in (+ Sf 24 for triangles / polygons, or for every face with QGD_SGEO=0: quadrilaterals rebuild Sf from their staged
# vertices), ~10 B of label lists (130 cells + 176 vertices per 128 faces on a box), 40 B out; per cell RecA 48 + RecB 32 +
# centre 24 = 104 B; per vertex RecA 48 + coordinates 24 = 72 B.  Reported beside the SURVEY figure, never instead of it.
OWN_BYTES_PER_FACE = 107 if os.environ.get("QGD_SGEO") == "0" else 83   # the bench box has quadrilateral faces only
OWN_BYTES_PER_CELL = 104
OWN_BYTES_PER_POINT = 72
# the fused one-launch step (fusedFaceCellKernel): SURVEY 8(d)'s three rows minus the bytes they hand each other through HBM (see main())
FUSED_BYTES_PER_FACE = 144
FUSED_BYTES_PER_CELL = 48 + 120
FUSED_BYTES_PER_POINT = 100
# whole explicit step, per cell-step on a hex box: vertex interp 196 + face kernel 648 + cell update 240
STEP_BYTES_PER_CELL = 1084
POINT_BYTES_PER_CELL = 196
CELL_BYTES_PER_CELL = 240


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--edge", dest="n", type=int, default=int(os.environ.get("QGD_BENCH_N", "400")), help="box edge in cells (n^3 cells in total)")
    ap.add_argument("--workload", default="qgd", choices=["qgd", "qhd", "implicit", "adjust"],
                    help="qgd: the headline (QGDFoam explicit step, BASELINE.json configs 3/4); qhd: the QHDFoam step of config 5 "
                         "(semi-implicit: flux assembly + pressure equation), its own metric line; implicit: the QGDFoam step with "
                         "implicitDiffusion true (the reference's default branch), one GPU, its own metric line; adjust: the QGDFoam explicit step "
                         "under Courant-number control (adjustTimeStep, QGDFoam.C L118-120), one GPU, its own metric line")
    ap.add_argument("--irregular", action="store_true", help="qhd: the config-5 stand-in mesh (jittered vertices, every 7th quad split into "
                                                             "triangles, labels shuffled in chunks then Morton-ordered) instead of a uniform box")
    ap.add_argument("--implicit-diffusion", action="store_true",
                    help="qhd: implicitDiffusion true, the reference's default [QGDThermo.C L70-82]: fvm::laplacian in the U and T equations, the four "
                         "systems as one multi-right-hand-side solve (one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the second line (fvsc drop-in path with host fields)")
    ap.add_argument("--dropin-n", type=int, default=200, help="box edge of the fvsc drop-in measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = debugging aid: all ranks share GPU 0 and halo messages are staged through host memory")
    ap.add_argument("--halo", default=os.environ.get("QGD_BENCH_HALO", "torch"), choices=["torch", "native"],
                    help="halo transport: torch.distributed P2P (default) or the library's own RCCL path (qgd_case_step_sharded)")
    ap.add_argument("--no-native-line", action="store_true",
                    help="N > 1: skip the second set of ranks that repeats the run over the library's own RCCL transport (native_transport)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip the two secondary lines (QHDFoam step and implicitDiffusion step at 200^3) appended under `secondary`")
    ap.add_argument("--check", action="store_true", help="print a checksum of the owned cells (to compare runs at different N)")
    ap.add_argument("--cpu-n", type=int, default=64, help="edge of each CPU-baseline rank's sample box")
    ap.add_argument("--cpu-steps", type=int, default=24)
    ap.add_argument("--cpu-only", action="store_true", help="only time the CPU baseline (no GPU needed) and print it")
    ap.add_argument("--cpu-ranks", type=int, default=0, help="CPU-baseline ranks (0 = one per physical core, at most 128)")
    return ap.parse_args()


def _cpu_rank_worker(n, steps, sync_dir, idx):
    """One single-threaded "rank" of the CPU baseline (child process started by cpu_baseline): the oracle on its own
    n^3 box.  Ranks start their timed region together through a ready/go file handshake."""
    import qgdsolver_amd as q
    from qgdsolver_amd.synthetic import box_initial_fields
    sys.path.insert(0, os.path.join(ROOT, "tests"))   # tests/oracle.py, the ctypes wrapper of oracle/: the cpu_baseline leg is the one part of
    from oracle import OracleCase, OracleMesh           # this file that may touch the checker

    mesh = q.PolyMesh.box(n, n, n)
    om = OracleMesh(mesh.primitives())
    h = 1.0 / n
    opt = q.default_options(stencil="GaussVolPoint", deltaT=0.1 * h / 1.3)
    oc = OracleCase(om, opt)
    U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
    oc.set_fields(U, T, p)
    oc.step(1)

    def barrier(tag):
        """all ranks enter a timed region together: ready file, then wait for the parent's go file"""
        open(os.path.join(sync_dir, f"ready{tag}_{idx}"), "w").close()
        go = os.path.join(sync_dir, f"go{tag}")
        t_wait = time.time()
        while not os.path.exists(go):
            if time.time() - t_wait > 600:
                sys.exit(3)
            time.sleep(0.005)

    barrier("a")
    t0 = time.perf_counter()
    oc.step(steps)
    print(f"CPU_RANK_SECONDS {time.perf_counter() - t0:.6f}", flush=True)
    # the same steps with the flux assembly fused (one vertex pass + one face pass, seven face fields stored: oracle/qgd_oracle.cpp
    # updateFluxesFused, same arithmetic -- tests/test_oracle_fused.py): what a CPU code organised for speed does with these formulas
    barrier("f")
    t0 = time.perf_counter()
    ok = oc.step_fused(steps)
    print(f"CPU_RANK_FUSED_SECONDS {(time.perf_counter() - t0) if ok else -1.0:.6f}", flush=True)
    # host-bandwidth yardstick on the same cores, all ranks at once: STREAM triad over 3 x 256 MB per rank
    import oracle as orc
    m = 32 * 1024 * 1024
    a, b, c = np.zeros(m), np.ones(m), np.full(m, 2.0)
    orc.lib.orc_stream_triad(orc._d(a), orc._d(b), orc._d(c), 3.0, m, 1)
    barrier("t")
    reps = 10
    t0 = time.perf_counter()
    orc.lib.orc_stream_triad(orc._d(a), orc._d(b), orc._d(c), 3.0, m, reps)
    print(f"CPU_RANK_TRIAD_GBS {24.0 * m * reps / (time.perf_counter() - t0) / 1e9:.4f}", flush=True)


def cpu_rank_budget(n):
    """How many single-threaded ranks the host really gives this process: physical cores, clipped by the scheduler
    affinity, the cgroup CPU quota (the GPU boxes expose 256 hardware threads but a 16-CPU quota) and half of the free
    memory (a rank holds about 5.2 kB of oracle fields per cell)."""
    cores = max(1, (os.cpu_count() or 2) // 2)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            words = open(path).read().split()
            quota = float(words[0])
            period = float(words[1]) if len(words) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cores = min(cores, max(1, int(quota / period)))
        except (OSError, ValueError, IndexError):
            pass
    try:
        avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
        try:
            avail = min(avail, int(open("/sys/fs/cgroup/memory.max").read()) - int(open("/sys/fs/cgroup/memory.current").read()))
        except (OSError, ValueError):
            pass
        cores = min(cores, int(0.5 * avail / (5200.0 * n ** 3)))
    except (OSError, IndexError, ValueError):
        pass
    return max(1, min(128, cores))


def host_description():
    """what the box's host is, beside what this process may use of it: CPU model, sockets, physical cores, hardware threads (lscpu's
    facts from /proc/cpuinfo) and the cgroup quota -- SURVEY 8(d) asks for one rank per PHYSICAL core; a container quota below that is
    a property of the measurement, stated with it"""
    model, phys, threads = None, set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                threads += 1
            elif not line.strip() and pid is not None:
                phys.add((pid, cid)); pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return {"cpu_model": model, "sockets": len({p for p, _ in phys}) or None, "physical_cores": len(phys) or None,
            "hardware_threads": threads or (os.cpu_count() or None), "cpu_quota": host_cpu_quota()}


def cpu_baseline(n, steps, ranks):
    """The CPU oracle (a restatement of the reference listings: the reference itself cannot be built here) timed on
    the host cores the way the reference would run: R single-threaded ranks, one per core, each with its own slab of
    the box (n^3 cells per rank, no halo traffic -- an upper bound for an MPI run), on a bounded sample.  The ranks
    are child processes (this process has initialised the GPU and must not fork workers)."""
    import subprocess
    import tempfile

    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    with tempfile.TemporaryDirectory() as sync_dir:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-rank-worker", str(n), str(steps), sync_dir, str(i)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for i in range(ranks)]
        for tag, limit in (("a", 280), ("f", 900), ("t", 900)):
            t_wait = time.time()
            while sum(os.path.exists(os.path.join(sync_dir, f"ready{tag}_{i}")) for i in range(ranks)) < ranks:
                if any(pr.poll() not in (None, 0) for pr in procs) or time.time() - t_wait > limit:
                    if tag == "a":
                        for pr in procs:
                            pr.kill()
                        return {"value": None, "unit": "Mcell-steps/s", "cores": ranks, "kind": "port", "sample": "CPU baseline ranks failed to start"}
                    break
                time.sleep(0.05)
            open(os.path.join(sync_dir, f"go{tag}"), "w").close()
        times, fused, triad = [], [], []
        for pr in procs:
            out, _ = pr.communicate(timeout=900)
            times += [float(line.split()[1]) for line in out.splitlines() if line.startswith("CPU_RANK_SECONDS")]
            fused += [float(line.split()[1]) for line in out.splitlines() if line.startswith("CPU_RANK_FUSED_SECONDS")]
            triad += [float(line.split()[1]) for line in out.splitlines() if line.startswith("CPU_RANK_TRIAD_GBS")]
    if len(times) != ranks:
        return {"value": None, "unit": "Mcell-steps/s", "cores": ranks, "kind": "port", "sample": "CPU baseline ranks failed"}
    dt = max(times)
    host = host_description()
    return {"value": ranks * n ** 3 * steps / dt / 1e6, "unit": "Mcell-steps/s", "cores": ranks, "kind": "port",
            "sample": f"{ranks} single-threaded oracle ranks x {n}^3-cell box x {steps} steps (field-at-a-time restatement "
                      f"of the reference listings, no halo exchange; slowest rank {dt:.1f} s); host: {host['cpu_model']}, "
                      f"{host['sockets']} socket(s), {host['physical_cores']} physical cores, {host['hardware_threads']} hardware threads, "
                      f"of which this process may use {host['cpu_quota']} (cgroup quota / affinity): the ranks are what the quota and "
                      f"the memory allow, NOT one per physical core",
            "host": host,
            "single_rank_value": n ** 3 * steps / min(times) / 1e6,
            # the same ranks, the same steps, the flux assembly fused (measured, not modelled)
            "fused": ({"value": ranks * n ** 3 * steps / max(fused) / 1e6, "unit": "Mcell-steps/s",
                       "sample": f"the same {ranks} ranks x {steps} steps with one vertex pass + one face pass per step instead of ~50 "
                                 f"field-at-a-time passes (slowest rank {max(fused):.1f} s)"}
                      if len(fused) == ranks and min(fused) > 0 else None),
            # what ANY CPU code (a perfectly fused one included) could reach on these cores: the step's algorithmic
            # bytes per cell-step against the STREAM-triad bandwidth the same ranks sustain together
            "host_triad_GBs": sum(triad) if len(triad) == ranks else None,
            "fused_cpu_upper_bound": (sum(triad) * 1e9 / STEP_BYTES_PER_CELL / 1e6) if len(triad) == ranks else None}


def dropin_fvsc_line(q, n, reps):
    """The literal drop-in path of north_star: an unmodified updateFluxes.H calls fvsc::grad(U), grad(e), grad(rho), grad(p)
    per step [QGDFoam/updateFluxes.H L41-65], each handing HOST fields over and taking a host face field back
    (qgd_fvsc_grad_v / qgd_fvsc_grad_s).  Times, per step of four calls at edge n: end to end (PCIe both ways included),
    and the kernels alone against their own compulsory bytes.  Reported beside the headline, never as the headline."""
    import ctypes as C
    from qgdsolver_amd import fvsc
    from qgdsolver_amd import _lib as L

    mesh = q.PolyMesh.box(n, n, n)
    dev = q.Device(mesh, fv_schemes={"fvsc": {"default": "GaussVolPoint"}}, fused_tables=False)   # stateless operators only
    rng = np.random.default_rng(1)
    nc, nb, nif, npnt = mesh.nCells, mesh.nBoundaryFaces, mesh.nInternalFaces, mesh.nPoints
    fields = [q.volField("U", rng.standard_normal((nc, 3)), rng.standard_normal((nb, 3)))] + \
             [q.volField(name, rng.standard_normal(nc), rng.standard_normal(nb)) for name in ("e", "rho", "p")]
    ms = (C.c_double * 3)()
    # host face fields are allocated (and touched) once, like the surface fields an OpenFOAM solver keeps alive
    # [QGDFoam/createFaceFluxes.H L40-218]; every call goes straight to the C entry
    sid = C.c_int()
    L.check(L.lib.qgd_stencil_lookup(dev._h, b"GaussVolPoint", C.byref(sid)), "qgd_stencil_lookup")
    outs = [np.ones((mesh.nFaces, 3 * vf.ncomp)) for vf in fields]
    dp = lambda a: a.ctypes.data_as(L.c_double_p)  # noqa: E731

    def call(vf, out):
        fn = L.lib.qgd_fvsc_grad_v if vf.ncomp == 3 else L.lib.qgd_fvsc_grad_s
        L.check(fn(dev._h, sid.value, dp(vf.internal), dp(vf.boundary), dp(out)), "qgd_fvsc_grad")

    for vf, out in zip(fields, outs):          # first calls size the persistent workspace
        call(vf, out)
    wall, parts = [], np.zeros(3)
    for _ in range(reps):
        t0 = time.perf_counter()
        for vf, out in zip(fields, outs):
            call(vf, out)
            L.lib.qgd_device_op_times(dev._h, ms)
            parts += np.array(list(ms))
        wall.append(time.perf_counter() - t0)
    parts /= reps
    # compulsory bytes of the four kernels pairs (vertex interpolation + face gradient), own layout: per internal face labels 25 B
    # + result 24*ncomp B; per cell value 8*ncomp + centre 32 B; per vertex value written and read 16*ncomp + coordinates 32 B +
    # its gather list 96 B + 8 cell values 8*ncomp*... counted once per cell above
    kb = sum(nif * (25 + 24 * k) + nc * (8 * k + 32) + npnt * (16 * k + 32 + 96) for k in (3, 1, 1, 1))
    pcie = sum(8 * (nc * k + nb * k + mesh.nFaces * 3 * k) for k in (3, 1, 1, 1))
    out = {"workload": f"fvsc::grad(U)+grad(e)+grad(rho)+grad(p), GaussVolPoint, {n}^3 cells, host fields in / host face fields out",
           "ms_per_step_end_to_end": 1e3 * min(wall), "ms_host_to_device": parts[0], "ms_kernels": parts[1], "ms_device_to_host": parts[2],
           "pcie_bytes_per_step": pcie, "pcie_GBs": pcie / min(wall) / 1e9,
           "kernel_bytes_per_step": kb, "kernel_frac_of_hbm_peak": kb / (parts[1] * 1e-3) / 1e9 / HBM_PEAK_GBS if parts[1] else None,
           "Mcell_steps_per_s_end_to_end": nc / min(wall) / 1e6}
    dev.close()
    mesh.close()
    return out


def host_cpu_quota():
    """CPUs this process may really use: hardware threads clipped by the scheduler affinity and the cgroup quota."""
    cpus = os.cpu_count() or 1
    try:
        cpus = min(cpus, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            words = open(path).read().split()
            quota = float(words[0])
            period = float(words[1]) if len(words) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cpus = min(cpus, max(1, int(quota / period)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, cpus)


def qgd_env():
    """every QGD_* variable set in this process's environment: they select code paths inside the library and the bench, so the line
    records them (a stray variable on the GPU box would otherwise change what is measured without a trace)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("QGD_")}


def bench_deadline_s():
    try:
        return float(os.environ.get("QGD_BENCH_DEADLINE_S", "900"))
    except ValueError:
        return 900.0


def arm_watchdog(seconds, what):
    """A rank started by somebody else's launcher (torch.distributed.run) has no relay parent of ours to end a hung collective: a
    daemon timer ends THIS process instead (collectives and synchronize release the GIL).  Exit status 124 like timeout(1)."""
    import threading

    def fire():
        print(f"bench.py: {what} exceeded QGD_BENCH_DEADLINE_S = {seconds:.0f} s; giving up", file=sys.stderr, flush=True)
        for pr in list(LIVE_CHILDREN):   # the second generation (secondary lines, native-transport ranks) goes with us
            kill_group(pr)
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def launch_ranks(n_ranks, argv, deadline_s, cmd=None):
    """Start n_ranks rank processes (one per GPU) as plain children, wait for them under a deadline, return
    {"line": rank 0's JSON line or None, "rc": worst status, "timed_out": bool, "stderr": rank 0's last stderr lines}.
    A rank that dies takes the job with it (its peers would wait in the rendezvous or the first exchange for ever); when the
    deadline passes every child is killed.  The caller never touches the GPU through this function.  Host set-up of the library
    is OpenMP: every rank gets cpu_quota // N threads so that N ranks do not oversubscribe the cgroup.  Variables a surrounding
    torch.distributed.run set for ITS ranks are not handed down (TORCHELASTIC_USE_AGENT_STORE would make rank 0 look for the agent's
    store instead of opening its own)."""
    import socket
    import subprocess
    import tempfile

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    threads = max(1, host_cpu_quota() // n_ranks)
    base = {k: v for k, v in os.environ.items()
            if not (k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE",
                                                            "LOCAL_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "OMP_NUM_THREADS_SET_BY_TORCHRUN"))}
    procs = []
    err0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if "OMP_NUM_THREADS" not in os.environ or "TORCHELASTIC_RUN_ID" in os.environ:   # torchrun pins 1 thread per rank: not ours
            env["OMP_NUM_THREADS"] = str(threads)
        procs.append(subprocess.Popen((cmd or [sys.executable, os.path.abspath(__file__)]) + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=err0 if r == 0 else None, text=(r == 0), start_new_session=True))
        LIVE_CHILDREN.append(procs[-1])
    worst, line, timed_out = 0, None, False
    t_end = time.time() + deadline_s
    try:
        import threading
        box = {}
        reader = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()), daemon=True)
        reader.start()   # rank 0's stdout is drained while we wait, so a long line cannot block it
        while any(pr.poll() is None for pr in procs):
            if [pr for pr in procs if pr.poll() not in (None, 0)]:
                break
            if time.time() > t_end:
                timed_out = True
                break
            time.sleep(0.2)
        if timed_out or [pr for pr in procs if pr.poll() not in (None, 0)]:
            for pr in procs:
                kill_group(pr)     # the rank AND whatever it started (rank 0's secondary lines / second set of ranks)
        for pr in procs:
            rc = pr.wait()
            worst = max(worst, abs(rc)) if rc else worst
        reader.join(10)
        for ln in (box.get("out") or "").splitlines():
            if ln.startswith("{"):
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)
    finally:
        for pr in procs:
            if pr.poll() is None:
                kill_group(pr)
            if pr in LIVE_CHILDREN:
                LIVE_CHILDREN.remove(pr)
    err0.seek(0)
    tail = err0.read().splitlines()[-12:]
    err0.close()
    if timed_out:
        worst = 124
    return {"line": line, "rc": worst, "timed_out": timed_out, "stderr": tail}


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher around it: start N rank processes BEFORE this process imports torch or touches
    HIP, relay rank 0's JSON line and exit with the worst child status.  Nothing is re-exec'ed: the parent stays a relay that never
    initialises the GPU, and it holds the deadline (QGD_BENCH_DEADLINE_S, default 900 s): a hung collective ends with the children
    killed and status 124 instead of burning the caller's whole timeout."""
    deadline = bench_deadline_s()
    res = launch_ranks(n_ranks, sys.argv[1:], deadline)
    for ln in res["stderr"]:
        print(ln, file=sys.stderr)
    if res["timed_out"]:
        print(f"bench.py: the ranks did not finish within QGD_BENCH_DEADLINE_S = {deadline:.0f} s; killed", file=sys.stderr)
    worst = res["rc"]
    if res["line"] and not res["timed_out"]:
        print(res["line"], flush=True)
    elif worst == 0:
        worst = 1
    sys.exit(min(worst, 255))


def kill_group(proc):
    """end a child started with start_new_session=True TOGETHER with everything it started (its ranks, its CPU-baseline workers):
    killing only the direct child would orphan a second generation that keeps the GPUs and an RCCL rendezvous busy"""
    import signal
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError, OSError):
        pass
    try:
        proc.kill()
    except OSError:
        pass


LIVE_CHILDREN = []   # process groups this process started and has not reaped (arm_watchdog and the deadline paths kill them)


def child_line(argv, timeout_s):
    """one bench.py child process (a one-GPU line of another workload, started after this process has released its device
    memory): its parsed JSON line, or {"error": ...}.  A child in a session of its own, never an exec: this process has
    initialised the GPU; on timeout the whole group goes."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC_")}
    pr = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, start_new_session=True)
    LIVE_CHILDREN.append(pr)
    try:
        stdout, stderr = pr.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        kill_group(pr)
        pr.communicate()
        return {"error": f"no result within {timeout_s:.0f} s"}
    finally:
        if pr in LIVE_CHILDREN:
            LIVE_CHILDREN.remove(pr)
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    if pr.returncode != 0 or not lines:
        return {"error": f"status {pr.returncode}", "stderr": stderr.splitlines()[-6:]}
    return json.loads(lines[-1])


def under_profiler():
    """rocprofv3 (and friends) preload a tool library into every process of the tree: a child workload would run under counter collection
    too, multiply the run time and write its own traces.  The secondary lines and the second set of ranks are skipped then."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("HSA_TOOLS_LIB", "") + os.environ.get("ROCP_TOOL_LIB", "")
    return ("rocprof" in pre) or any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ)


def secondary_lines(args):
    """The two workloads the headline never touches (qgd_poisson.hip + qgd_qhd.hip, qgd_implicit.hip), 20 timed steps each after
    warm-up at 200^3, as child processes once the headline's device memory is released.  Same keys as the stand-alone lines
    (`--workload qhd`, `--workload implicit`), cut to what a reader compares."""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "phase_ms", "step_roofline_frac", "setup_s", "error", "stderr")
    out = {}
    n = os.environ.get("QGD_BENCH_SECONDARY_N", "200")   # tests shrink it; any value but 200 is visible in the key and in config.env
    # (20 warm-up steps like the headline -- SURVEY 8(d): "steady state: >= 20 warm-up steps discarded"; rounds 4-5 timed steps 6-25 / 11-30, in
    # which the iterative solves' start values were still building their history)
    lines = [(f"qhd_n{n}", ["--workload", "qhd", "--edge", n, "--steps", "20", "--warmup", "20"]),
             (f"qhd_implicit_n{n}", ["--workload", "qhd", "--implicit-diffusion", "--edge", n, "--steps", "20", "--warmup", "20"]),
             (f"implicit_n{n}", ["--workload", "implicit", "--edge", n, "--steps", "20", "--warmup", "20"]),
             (f"adjust_n{n}", ["--workload", "adjust", "--edge", n, "--steps", "50", "--warmup", "10"])]
    # BASELINE config 5 (16 M irregular cells, QHDFoam) on one GPU: ~45 s of host mesh set-up + 30 steps.  QGD_BENCH_C5=0 leaves it out,
    # QGD_BENCH_C5_N shrinks it (tests), QGD_BENCH_C5_IMPLICIT=1 adds the implicitDiffusion branch on the same mesh.
    if os.environ.get("QGD_BENCH_C5", "1") != "0":
        c5n = os.environ.get("QGD_BENCH_C5_N", "252")
        key = "qhd_c5" if c5n == "252" else f"qhd_c5_n{c5n}"
        lines.append((key, ["--workload", "qhd", "--irregular", "--edge", c5n, "--steps", "20", "--warmup", "20"]))
        if os.environ.get("QGD_BENCH_C5_IMPLICIT", "0") == "1":
            lines.append((key + "_implicit", ["--workload", "qhd", "--irregular", "--implicit-diffusion", "--edge", c5n, "--steps", "20", "--warmup", "20"]))
    for key, argv in lines:
        if bench_deadline_s() - (time.perf_counter() - T_START) < 150.0:
            out[key] = {"error": "not started: less than 150 s left before QGD_BENCH_DEADLINE_S"}
            continue
        t0 = time.perf_counter()
        d = child_line(argv, 420)
        out[key] = {k: d[k] for k in keep if k in d}
        out[key]["wall_s"] = time.perf_counter() - t0
    return out


def native_transport_line(args, world, t_budget_s):
    """N > 1: the same run again on a FRESH set of ranks whose halo transport is the library's own RCCL path
    (qgd_case_step_sharded -- what a C++/MPI host of the C-ABI uses; torch.distributed only carries the unique id and the timing
    barrier there, over gloo).  Returns what the first line is compared on, or {"error": ...}: the second set must never cost the
    first line."""
    argv = ["--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup), "--edge", str(args.n), "--backend", args.backend,
            "--halo", "native", "--no-native-line", "--no-secondary", "--no-cpu-baseline", "--no-dropin", "--check"]
    res = launch_ranks(world, argv, t_budget_s)
    if not res["line"] or res["rc"]:
        return {"error": ("no result within %.0f s" % t_budget_s) if res["timed_out"] else f"status {res['rc']}", "stderr": res["stderr"][-6:]}
    d = json.loads(res["line"])
    return {"ms_per_step": d["ms_per_step"], "value": d["value"], "checksum_rho": d.get("checksum_rho"), "transport": d["config"].get("transport"),
            "rccl_ranks": d["config"].get("rccl_ranks"), "overlap_selfcheck_rel": d.get("overlap_selfcheck_rel"),
            "partition": d["config"].get("partition")}


# ---- QHDFoam (config 5) -----------------------------------------------------------------------------------------------
# ALGORITHMIC bytes per cell-step of the QHDFoam step on a hexahedral mesh with GaussVolPoint (3 internal faces and 1 vertex per
# cell; face geometry counted as SURVEY.md 8(d) counts it for the QGD kernel: 80 B of Gauss coefficients per face), by pass:
#   vertex values of {U,T}: gather list 8 x (4 + 8) + 4, 32 read, 32 written                                   = 164
#   face pass 1 [updateFields.H L36-73, updateFluxes.H L33-38]: per face 8 + 24 + 8 + 16 + 80 + tau 8 in, phiu, phiwo,
#       phiTauTReg, Uf & gradUf (3) out = 48 (BdFrcf is formed again in pass 2); cell and vertex records 32 + 32 = 3 x 192 + 64 = 640
#   fvc::grad(U): 6 face labels 24, 3 x Sf 24, V 8, U 24, 72 written                                            = 200
#   vertex values of p: 100 + 8 + 8                                                                             = 116
#   face pass 2 [QHDUEqn.H L36-84, QHDTEqn.H L65-91]: per face 136 + tau, phi, phiTauTReg 24 + ugu 24 in, 32 out;
#       per cell {U,T} 32 + p 8 + grad(U) 72, per vertex p 8                                                    = 3 x 216 + 120 = 768
#   cell update: 24 + 3 x 32 + V 8 + 32 read + 32 written                                                       = 192
# and per iteration of the pressure solve [QHDpEqn.H L35-47] (multigrid-preconditioned CG; single-precision cycle):
#   level 0 of the V-cycle: 4 full sweeps x (6 x (4 + 4) + 16) = 256 (the last one reads r and writes z in double: + 12), the first
#   sweep from zero 12 (written by the CG's axpy kernel since round 4), the smoothed prolongator and its transpose (4.3 entries per
#   fine row each, x 8, + 8 and 4 for the vectors) 82;  level 1 (1/8 of the rows, 34.5 entries per row):
#   4 x (34.5 x 8 + 16) / 8 = 146;  the levels below ~10                                                       = 518
#   CG: A d 6 x 12 + 24 = 96, x and r updates 40 (|r| summed on the way; r.z is summed by the cycle's last sweep), new direction 24 = 160
#   (round 3, separate passes: + two dot products 32 and the precision conversions 24 - 12 = 722 in all)
QHD_EXPLICIT_BYTES_PER_CELL = 164 + 640 + 200 + 116 + 768 + 192
QHD_BYTES_PER_CELL_PER_ITERATION = 518 + 160


def qhd_line(args):
    """python bench.py --workload qhd [--edge N] [--irregular]: Mcell-steps/s of the resident QHDFoam step (qgd_qhd_case_step)."""
    import qgdsolver_amd as q
    from qgdsolver_amd import qhdfoam

    if q.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: qgdsolver_amd has no CPU fallback")
    n = args.n
    t_setup = time.perf_counter()
    if args.irregular:
        from qgdsolver_amd.synthetic import c5_mesh
        mesh = c5_mesh(n, 64 ** 3)
    else:
        mesh = q.PolyMesh.box(n, n, n)
    h = 1.0 / n
    # QGD_QHD_FUSED=1: the cell blocks of QGDFoam's one-launch step carry the explicit branch's U and T equations (qgd_qhd.hip; off by default --
    # no gain measured); otherwise their tables are not this workload's
    blocks = (not args.implicit_diffusion) and os.environ.get("QGD_QHD_FUSED", "0") == "1"
    dev = q.Device(mesh, fused_tables=True if blocks else False)
    opt = qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=0.1, rho0=1.0, mu=1e-3, Pr=0.71, beta=3.4e-3,
                              g=(0.0, -9.81, 0.0), deltaT=0.02 * h / 0.1, pTol=1e-8, pMaxIter=300, pRefCell=0,
                              implicitDiffusion=1 if args.implicit_diffusion else 0, implicitTol=1e-10, implicitMaxIter=1000)
    case = qhdfoam.QHDFoamCase(dev, opt)
    WALL = dict(U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("qhdFluxCoupled", None))
    for ip in range(mesh.nPatches):   # buoyant cavity: hot xMin, cold xMax, adiabatic walls, impermeable (qhdFlux fed by the flux)
        case.set_bc(ip, U=WALL["U"], T=("fixedValue", 310.0) if ip == 0 else (("fixedValue", 290.0) if ip == 1 else WALL["T"]), p=WALL["p"])
    C = mesh.array("C").reshape(-1, 3)
    rng = np.random.default_rng(5)
    U = np.zeros((mesh.nCells, 3))
    U[:, 0] = 0.1 * np.sin(np.pi * C[:, 0]) * np.cos(np.pi * C[:, 1])
    U[:, 1] = -0.1 * np.cos(np.pi * C[:, 0]) * np.sin(np.pi * C[:, 1])
    T = 300.0 + 10.0 * (0.5 - C[:, 0]) + 0.1 * rng.standard_normal(mesh.nCells)
    case.set_fields(U, T, np.zeros(mesh.nCells))
    del U, T, C
    nc = mesh.nCells
    t_setup = time.perf_counter() - t_setup
    case.step(max(args.warmup, 1))
    t0 = time.perf_counter()
    case.step(args.steps)          # returns after the device has finished
    elapsed = time.perf_counter() - t0
    info = case.info()
    # second pass: where the time goes (host clock around stream-ordered phases, each followed by a wait)
    phase_ms = {"assemble": 0.0, "solve": 0.0, "advance": 0.0}
    iters = []
    reps = 0 if args.implicit_diffusion else min(args.steps, 5)   # (the implicit branch's advance is phases 7, 10..16: timed as a whole step only)
    for _ in range(reps):
        case.sync(); t = time.perf_counter()
        case.step_phase(0); case.sync()
        phase_ms["assemble"] += time.perf_counter() - t; t = time.perf_counter()
        for k in (1, 2):
            case.step_phase(k)
        while not case.solve_status()["done"]:
            for k in (3, 4, 5):
                case.step_phase(k)
        iters.append(case.solve_status()["iterations"])
        phase_ms["solve"] += time.perf_counter() - t; t = time.perf_counter()
        for k in (6, 7, 8):
            case.step_phase(k)
        case.sync()
        phase_ms["advance"] += time.perf_counter() - t
    phase_ms = {k: 1e3 * v / reps for k, v in phase_ms.items()} if reps else None
    impl = case.implicit_info() if args.implicit_diffusion else None
    sw = case.sweep_time(30)
    sweep_bytes = sw["rows"] * (sw["width"] * (4 + sw["value_bytes"]) + 4 * sw["value_bytes"])
    achieved = sweep_bytes / (sw["ms"] * 1e-3) / 1e9 if sw["ms"] else None
    it = info["pIterations"]
    step_bytes = nc * (QHD_EXPLICIT_BYTES_PER_CELL + QHD_BYTES_PER_CELL_PER_ITERATION * it)
    if impl:   # the four-component Chebyshev step: lists 48 + face coefficients 24 + per component diag, rhs, x, d read and d, x written 48
        step_bytes += nc * (48 + 24 + 4 * 48) * max(v["iterations"] for v in impl["solves"].values())
    out = {
        "metric": "Mcell-steps/s (QHDFoam step)", "value": nc * args.steps / elapsed / 1e6, "unit": "Mcell-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"QHDFoam buoyant cavity, {nc / 1e6:.1f}M cells, "
                                + ("config-5 stand-in mesh (jittered hexahedra, every 7th quad split into triangles, Morton order)" if args.irregular
                                   else "uniform hex box (blockMesh numbering)")
                                + ", GaussVolPoint, HbyUQHD, pressure equation to 1e-8 with multigrid-preconditioned CG"
                                + (", implicitDiffusion true (the reference's default): U and T systems as four right-hand sides of one "
                                   f"{impl['solver']} solve to 1e-10" if impl else ", implicitDiffusion false")),
                   "cells": nc, "pressure_iterations_per_step": it, "multigrid_levels": info["mgLevels"],
                   "implicit_iterations": {k: v["iterations"] for k, v in impl["solves"].items()} if impl else None,
                   "implicit_unconverged_steps": impl["unconverged_steps"] if impl else None,
                   "implicit_stalled_steps": impl["stalled_steps"] if impl else None, "fused_step": case.fused_info(), "env": qgd_env()},
        "roofline": {"bound": "hbm", "kernel": "mgSmoothKernel<float>, multigrid level 0 (one damped-Jacobi sweep of the pressure preconditioner)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
                     "traffic": (secondary_traffic("qhd_n200") if (n == 200 and not args.irregular and not args.implicit_diffusion)
                                 else (secondary_traffic("qhd_c5") if (n == 252 and args.irregular and not args.implicit_diffusion) else None)),
                     "traffic_is_static": True,
                     "traffic_source": SECONDARY_TRAFFIC_SOURCE, "algorithmic_bytes_per_launch": sweep_bytes, "avg_launch_ms": sw["ms"]},
        "step_bytes_model": {"explicit_bytes_per_cell": QHD_EXPLICIT_BYTES_PER_CELL, "bytes_per_cell_per_pressure_iteration": QHD_BYTES_PER_CELL_PER_ITERATION,
                             "bytes_per_step": step_bytes},
        "step_roofline_frac": step_bytes * args.steps / elapsed / (HBM_PEAK_GBS * 1e9),
        # which kernel the step spends most of its GPU time in: from the committed profile of this workload; when it is
        # the multigrid sweep, `roofline` above is that kernel's level-0 instance measured live
        "largest_kernel": (largest_kernel(f"r06_qhd_n{n}_kernel_stats.csv") if (not args.irregular and not args.implicit_diffusion) else None),
        "phase_ms": phase_ms, "pressure_iterations_second_pass": iters, "pressure_final_residual": info["pFinalResidual"],
        "setup_s": t_setup,
    }
    case.close(); dev.close()
    print(json.dumps(out), flush=True)


# the implicitDiffusion branch: bytes one Jacobi-PCG iteration of the three-component U system moves per cell in the build's layout
# (hexahedra: 6 list entries x 8 B + 3 face coefficients x 8 B (each face serves two cells) + diag, direction, product 3 x 8 B each
# = 144 B for the matrix product; x, r, d, q, diag read and x, r written = 168 B for the update; r, diag, d read and d written =
# 96 B for the direction), and a third of the vector part for the one-component e system
IMPL_APPLY_BYTES_PER_CELL = 144
# one Chebyshev step of the three-component system (round 4's default solver: product, d, next iterate, partial residual sums in ONE
# kernel): 6 list entries x 8 B + 3 face coefficients x 8 B + per component diag, rhs, x, d read and d, x written = 3 x 48 B
IMPL_CHEB_BYTES_PER_CELL = 48 + 24 + 3 * 48
IMPL_CHEB_BYTES_PER_CELL_E = 48 + 24 + 48
IMPL_ITER_BYTES_PER_CELL_U = 144 + 168 + 96
IMPL_ITER_BYTES_PER_CELL_E = 48 + 24 + 24 + 56 + 32


def largest_kernel(stats_csv):
    """the kernel with the largest share of the GPU time of a workload, from the committed rocprofv3 --kernel-trace --stats summary of the
    builder's profiling run of that workload (static, like `traffic`): {"kernel", "share", "avg_us", "calls", "source"} or None"""
    path = os.path.join(ROOT, "profiles", stats_csv)
    try:
        import csv
        rows = list(csv.DictReader(open(path)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        r = max(rows, key=lambda x: float(x["TotalDurationNs"]))
        name = r["Name"].replace("qgd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        return {"kernel": name, "share": float(r["TotalDurationNs"]) / tot, "avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]),
                "source": "profiles/" + stats_csv, "is_static": True}
    except Exception:
        return None


def secondary_traffic(key):
    """HBM-side bytes per launch of the kernel in a secondary line's `roofline` object: a constant from profiles/r03_pmc_secondary.json
    (TCC EA request counters of the builder's profiling run of this workload at 200^3, scripts/collect_secondary_pmc.sh), not counters
    of THIS run; None when the file or the key is missing (other sizes, the irregular mesh)."""
    try:
        with open(os.path.join(ROOT, "profiles", SECONDARY_TRAFFIC_FILE)) as f:
            return json.load(f)[key]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


SECONDARY_TRAFFIC_FILE = "r05_pmc_secondary.json"
SECONDARY_TRAFFIC_SOURCE = ("rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ{,_64B}_sum in separate passes over "
                            "scripts/secondary_kernel_probe.py (scripts/collect_secondary_pmc.sh), bytes = sum(size*requests), average of "
                            "the 30 launches of the measurement entry; L2-miss traffic, Infinity-Cache hits included; profiles/" + SECONDARY_TRAFFIC_FILE)


def implicit_line(args):
    """python bench.py --workload implicit [--edge N]: Mcell-steps/s of the QGDFoam step with implicitDiffusion true, the reference's
    default branch [QGDThermo.C L70-82]: the explicit flux assembly without the viscous parts + the U and e systems by Jacobi-PCG."""
    import qgdsolver_amd as q
    from qgdsolver_amd.synthetic import box_initial_fields

    if q.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: qgdsolver_amd has no CPU fallback")
    n = args.n
    t_setup = time.perf_counter()
    mesh = q.PolyMesh.box(n, n, n)
    dev = q.Device(mesh)   # (with the cell blocks: the branch assembles its U systems on them, QGD_IMPL_FUSED)
    opt = q.default_options(stencil="GaussVolPoint", deltaT=0.1 / n / 1.3, implicitDiffusion=1, mu=1e-3)
    case = q.QGDFoamCase(dev, opt)
    U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    del U, T, p
    nc = mesh.nCells
    t_setup = time.perf_counter() - t_setup
    case.step(max(args.warmup, 1))
    t0 = time.perf_counter()
    case.step(args.steps)          # returns after the device has finished
    elapsed = time.perf_counter() - t0
    info, solves = case.info(), case.implicit_info()
    fused_impl = case.fused_info().get("fusedImplicit", False)
    ap = case.implicit_apply_time(30)
    cheb = solves.get("solver") == "chebyshev"
    apply_bytes = (IMPL_CHEB_BYTES_PER_CELL if cheb else IMPL_APPLY_BYTES_PER_CELL) * ap["rows"]
    achieved = apply_bytes / (ap["ms"] * 1e-3) / 1e9 if ap["ms"] else None
    it_u = max(solves["solves"][k]["iterations"] for k in ("Ux", "Uy", "Uz"))
    it_e = solves["solves"]["e"]["iterations"]
    out = {
        "metric": "Mcell-steps/s (QGDFoam step, implicitDiffusion true)", "value": nc * args.steps / elapsed / 1e6, "unit": "Mcell-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"QGDFoam {n}^3 = {nc / 1e6:.1f}M-cell uniform hex box (blockMesh numbering), GaussVolPoint, constScPrModel1, "
                                "implicitDiffusion true (the reference's default), mu = 1e-3, zeroGradient patches, fixed deltaT, "
                                + ("U and e systems by Chebyshev iteration on the Jacobi-preconditioned systems to 1e-10" if cheb
                                   else "U and e systems by Jacobi-PCG to 1e-10 (QGD_IMPL_SOLVER=pcg)")),
                   "cells": nc, "iterations_U": it_u, "iterations_e": it_e, "unconverged_steps": solves["unconverged_steps"], "stalled_steps": solves["stalled_steps"],
                   # True: vertex values, QGD fluxes, tauMC and the rows of the three U systems are ONE launch on the fused step's cell blocks
                   "fused_assembly_of_the_U_systems": bool(fused_impl),
                   "env": qgd_env()},
        "roofline": {"bound": "hbm", "kernel": ("iChebKernel<3,0> (one Chebyshev step of the three-component U system: matrix product, d, next iterate and the "
                                                "partial residual sums in one walk of the matrix for the three right-hand sides)" if cheb else
                                                "iApplyKernel<3,1> (matrix product of the three-component U system, one walk of the matrix for the three right-hand sides)"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
                     "traffic": secondary_traffic("implicit_n200") if (n == 200 and cheb) else None, "traffic_is_static": True,
                     "traffic_source": SECONDARY_TRAFFIC_SOURCE, "algorithmic_bytes_per_launch": apply_bytes, "avg_launch_ms": ap["ms"]},
        "solver_bytes_model": {"per_cell_per_iteration_U": IMPL_CHEB_BYTES_PER_CELL if cheb else IMPL_ITER_BYTES_PER_CELL_U,
                               "per_cell_per_iteration_e": IMPL_CHEB_BYTES_PER_CELL_E if cheb else IMPL_ITER_BYTES_PER_CELL_E,
                               "bytes_per_step_in_the_solves": nc * ((IMPL_CHEB_BYTES_PER_CELL if cheb else IMPL_ITER_BYTES_PER_CELL_U) * it_u
                                                                     + (IMPL_CHEB_BYTES_PER_CELL_E if cheb else IMPL_ITER_BYTES_PER_CELL_E) * it_e)},
        "largest_kernel": largest_kernel(f"r06_implicit_n{n}_kernel_stats.csv"),
        "min_rho": info["minRho"], "setup_s": t_setup,
    }
    case.close(); dev.close()
    print(json.dumps(out), flush=True)


def adjust_line(args):
    """python bench.py --workload adjust [--edge N]: Mcell-steps/s of the QGDFoam explicit step under Courant-number control (adjustTimeStep:
    QGDCourantNo.H L36-53 + setDeltaT-QGDQHD.H L41-61 every step [QGDFoam.C L118-120]): every face's Courant number and tauQGDf before the first
    cell may advance -- the cell blocks up to their flux sums and Courant partials, the two-level reduction, deltaT on the device, the cell kernel."""
    import qgdsolver_amd as q
    from qgdsolver_amd.synthetic import box_initial_fields

    if q.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: qgdsolver_amd has no CPU fallback")
    n = args.n
    t_setup = time.perf_counter()
    mesh = q.PolyMesh.box(n, n, n)
    dev = q.Device(mesh)
    opt = q.default_options(stencil="GaussVolPoint", deltaT=0.05 / n / 1.3, adjustTimeStep=1, maxCo=0.1, maxDeltaT=1.0)
    case = q.QGDFoamCase(dev, opt)
    U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
    case.set_fields(U, T, p)
    del U, T, p
    nc = mesh.nCells
    t_setup = time.perf_counter() - t_setup
    fi = case.fused_info()
    case.step(max(args.warmup, 1))
    t0 = time.perf_counter()
    case.step(args.steps)          # returns after the device has finished
    elapsed = time.perf_counter() - t0
    info = case.info()
    out = {
        "metric": "Mcell-steps/s (QGDFoam explicit step, adjustTimeStep)", "value": nc * args.steps / elapsed / 1e6, "unit": "Mcell-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"QGDFoam {n}^3 = {nc / 1e6:.1f}M-cell uniform hex box (blockMesh numbering), GaussVolPoint, constScPrModel1, explicit "
                                "diffusion, zeroGradient patches, adjustTimeStep with maxCo 0.1 (deltaT recomputed on the device every step)"),
                   "cells": nc, "deltaT": info["deltaT"], "CoNum": info["CoNum"],
                   "path": ("cell blocks in two launches (fusedFaceCellKernel<..., ADJ> up to the flux sums and Courant partials, cellFinishKernel)"
                            if fi.get("fusedAdjust") else "three kernels (vertex values, faces with the Courant partials, cells)"),
                   "blocks": fi["blocks"], "env": qgd_env()},
        "step_roofline_frac": STEP_BYTES_PER_CELL * nc * args.steps / elapsed / (HBM_PEAK_GBS * 1e9),
        "min_rho": info["minRho"], "setup_s": t_setup,
    }
    case.close(); dev.close()
    print(json.dumps(out), flush=True)


def qhd_line_sharded(args):
    """python bench.py --workload qhd --gpus N [--irregular]: the QHDFoam step on N cell-range shards, one rank per GPU -- config 5 as
    configured.  Transport: the library's own RCCL path (qgd_qhd_case_step_sharded: halo messages, all-reduced PCG scalars, the comm
    points of the multigrid hierarchy that spans the ranks); with --backend gloo every rank sits on GPU 0 and DistWorld stages the
    messages through host tensors (the debugging mode a 1-GPU box can run)."""
    import ctypes as C
    import torch
    import torch.distributed as dist

    import qgdsolver_amd as q
    from qgdsolver_amd import _lib as L, qhdfoam
    from qgdsolver_amd.halo import DistWorld, NativeComm, QhdStepper, slab_range

    world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    staged = args.backend == "gloo"
    if staged:
        local_rank = 0
    if q.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: qgdsolver_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # torch.distributed is the control plane only (unique id, timing barrier, checksum) and runs over gloo on the host: the data path
    # is the library's own RCCL communicator, and no second RCCL instance is initialised in the process
    dist.init_process_group(backend="gloo")
    n = args.n
    t_setup = time.perf_counter()
    if args.irregular:
        from qgdsolver_amd.synthetic import c5_mesh
        g = c5_mesh(n, 64 ** 3)
        mesh = g.shard(world, rank)
        n_global = g.nCells
        cg = mesh.array("cellGlobal")
        lo, hi = (n_global * rank) // world, (n_global * (rank + 1)) // world
        owned = int(((cg >= lo) & (cg < hi)).sum())
        peers = [int(p) for p in mesh.array("haloPeer")]
        g.close()
    else:
        lo, hi, k_lo, k_hi = slab_range(n, rank, world)
        mesh = q.PolyMesh.box(n, n, n, k_range=(k_lo, k_hi))
        n_global = n ** 3
        cg = np.arange(n * n * k_lo, n * n * k_hi)
        owned = n * n * (hi - lo)
        peers = [rank - 1 if rank > 0 else -1, rank + 1 if rank < world - 1 else -1]
    h = 1.0 / n
    dev = q.Device(mesh, device_id=local_rank, fused_tables=False)
    opt = qhdfoam.qhd_options(stencil="GaussVolPoint", tauModel="HbyUQHD", aQGD=0.5, UQHD=0.1, rho0=1.0, mu=1e-3, Pr=0.71, beta=3.4e-3,
                              g=(0.0, -9.81, 0.0), deltaT=0.02 * h / 0.1, pTol=1e-8, pMaxIter=300, pRefCell=0)
    case = qhdfoam.QHDFoamCase(dev, opt)
    WALL = dict(U=("fixedValue", (0.0, 0.0, 0.0)), T=("zeroGradient", None), p=("qhdFluxCoupled", None))
    for ip in range(mesh.nPatches):
        case.set_bc(ip, U=WALL["U"], T=("fixedValue", 310.0) if ip == 0 else (("fixedValue", 290.0) if ip == 1 else WALL["T"]), p=WALL["p"])
    Cc = mesh.array("C").reshape(-1, 3)
    noise = np.random.default_rng(5).standard_normal(n_global)[cg]      # a function of the GLOBAL cell label: shards agree with the whole
    U = np.zeros((mesh.nCells, 3))
    U[:, 0] = 0.1 * np.sin(np.pi * Cc[:, 0]) * np.cos(np.pi * Cc[:, 1])
    U[:, 1] = -0.1 * np.cos(np.pi * Cc[:, 0]) * np.sin(np.pi * Cc[:, 1])
    case.set_fields(U, 300.0 + 10.0 * (0.5 - Cc[:, 0]) + 0.1 * noise, np.zeros(mesh.nCells))
    del U, Cc, noise
    t_setup = time.perf_counter() - t_setup
    if staged:
        def to_t(ptr, cnt):
            return torch.from_numpy(dev.to_host(ptr, (int(cnt),)))

        def from_t(t, ptr):
            a = np.ascontiguousarray(t.numpy())
            if a.nbytes:
                L.check(L.lib.qgd_device_copy(dev._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "qgd_device_copy")

        stepper = QhdStepper(DistWorld(case, dist, torch, peers, to_t, from_t))
        run = stepper.step
        transport = "gloo, host-staged (debugging mode: every rank on GPU 0)"
    else:
        def bcast(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        comm = NativeComm(local_rank, rank, world, bcast=bcast)
        comm.qhd_exchange(case, peers, 0)          # ghost cells start from their owners
        run = lambda k: comm.qhd_step(case, peers, k)   # noqa: E731
        transport = "RCCL inside the library (qgd_qhd_case_step_sharded)"
    t_first = time.perf_counter()
    run(max(args.warmup, 1))                       # the first step also builds the multigrid hierarchy that spans the ranks
    case.sync()
    t_first = time.perf_counter() - t_first
    dist.barrier()
    t0 = time.perf_counter()
    run(args.steps)
    case.sync()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    info = case.info()
    check = torch.tensor([float(np.abs(case.field("p")[: min(owned, mesh.nCells)]).sum())], dtype=torch.float64)
    dist.all_reduce(check, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({
            "metric": "Mcell-steps/s (QHDFoam step)", "value": n_global * args.steps / elapsed / 1e6, "unit": "Mcell-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"QHDFoam buoyant cavity, {n_global / 1e6:.1f}M cells on {world} cell-range shards, "
                                    + ("config-5 stand-in mesh (jittered hexahedra, every 7th quad split into triangles, Morton order)" if args.irregular
                                       else "uniform hex box cut into k-slabs")
                                    + ", GaussVolPoint, HbyUQHD, pressure equation to 1e-8, multigrid hierarchy spanning the ranks"),
                       "cells": n_global, "cells_per_gpu": n_global // world, "transport": transport,
                       "pressure_iterations_per_step": info["pIterations"], "multigrid_levels": info["mgLevels"],
                       "rccl_ranks": None if staged else comm.info()["ranks"], "env": qgd_env()},
            "roofline": {"bound": "hbm", "kernel": "distributed level 0 of the multigrid cycle: no separate sweep timing on shards", "achieved": None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
            "pressure_final_residual": info["pFinalResidual"], "first_steps_s": t_first, "setup_s": t_setup,
            "sum_abs_p_owned_prefix": float(check.item())}), flush=True)
    case.close(); dev.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.workload == "implicit":
        if "QGD_BENCH_N" not in os.environ and "--edge" not in " ".join(sys.argv):
            args.n = 200
        if args.gpus > 1:
            print("bench.py: --workload implicit is a one-GPU line (the branch shards: tests/test_implicit_sharded.py)", file=sys.stderr)
            sys.exit(2)
        implicit_line(args)
        return
    if args.workload == "adjust":
        if "QGD_BENCH_N" not in os.environ and "--edge" not in " ".join(sys.argv):
            args.n = 200
        if args.gpus > 1:
            print("bench.py: --workload adjust is a one-GPU line (the all-reduce of the Courant number on shards: tests/test_halo_gloo.py)", file=sys.stderr)
            sys.exit(2)
        adjust_line(args)
        return
    if args.workload == "qhd":
        if "QGD_BENCH_N" not in os.environ and "--edge" not in " ".join(sys.argv):
            args.n = 252 if args.irregular else 200
        if args.gpus > 1 and args.implicit_diffusion:
            print("bench.py: --workload qhd --implicit-diffusion is a one-GPU line (the branch shards: tests/test_qhd_implicit.py)", file=sys.stderr)
            sys.exit(2)
        if args.gpus > 1:
            if "WORLD_SIZE" not in os.environ:
                self_launch(args.gpus)   # never returns
            if int(os.environ["WORLD_SIZE"]) != args.gpus:
                print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher and the flag disagree", file=sys.stderr)
                sys.exit(2)
            qhd_line_sharded(args)
            return
        qhd_line(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.cpu_only:
        self_launch(args.gpus)   # never returns
    if "TORCHELASTIC_RUN_ID" in os.environ and os.environ.get("OMP_NUM_THREADS") == "1":
        # torch.distributed.run pins one OpenMP thread per rank; the library's host set-up (mesh tables) is OpenMP and not timed
        os.environ["OMP_NUM_THREADS"] = str(max(1, host_cpu_quota() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        arm_watchdog(bench_deadline_s(), f"rank {os.environ.get('RANK', '?')}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)

    if args.cpu_only:
        print(json.dumps(cpu_baseline(args.cpu_n, args.cpu_steps, args.cpu_ranks or cpu_rank_budget(args.cpu_n))))
        return

    import torch
    import torch.distributed as dist

    import qgdsolver_amd as q
    from qgdsolver_amd import _lib as L
    from qgdsolver_amd.halo import SlabHalo, slab_range
    from qgdsolver_amd.synthetic import box_initial_fields

    if not torch.cuda.is_available() or q.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: qgdsolver_amd has no CPU fallback")
    staged = args.backend == "gloo"
    if staged:
        local_rank = 0  # every rank on GPU 0; exercises the multi-rank logic on a 1-GPU box
    # --halo native: the DATA path is the library's own RCCL communicator; torch.distributed only carries the unique id, the timing
    # barrier and the checksum, over gloo on the host, so that no second RCCL instance is initialised in the process
    ctrl_host = staged or (args.halo == "native" and world > 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctrl_host:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n = args.n
    lo, hi, k_lo, k_hi = slab_range(n, rank, world)
    t_setup = time.perf_counter()
    mesh = q.PolyMesh.box(n, n, n, k_range=(k_lo, k_hi))
    owned_cells = n * n * (hi - lo)
    dev = q.Device(mesh, device_id=local_rank)
    h = 1.0 / n
    opt = q.default_options(stencil="GaussVolPoint", deltaT=0.1 * h / 1.3)  # fixed deltaT = 0.1 h / (c+|U|)_max
    case = q.QGDFoamCase(dev, opt)
    C = mesh.array("C").reshape(-1, 3)
    U, T, p = box_initial_fields(C)
    # the noise must be a function of the GLOBAL cell label so that shards agree with the unsharded run
    rng = np.random.Generator(np.random.MT19937(12345))
    noise = rng.uniform(-1e-3, 1e-3, size=n * n * n)
    T = 1.0 + noise[n * n * k_lo: n * n * k_hi]
    del noise
    case.set_fields(U, T, p)
    init_fields = (U, T, p) if world > 1 else None
    del U, T, p, C
    n_if, n_c, n_p = mesh.nInternalFaces, mesh.nCells, mesh.nPoints
    mesh.close()
    stream = torch.cuda.current_stream()
    case.set_stream(stream.cuda_stream)
    t_setup = time.perf_counter() - t_setup

    # one RCCL send/recv pair per neighbour per step; buffers live in HBM, pack/unpack run on the same stream
    if not staged:
        halo = SlabHalo(case, rank, world, dist, alloc=lambda c: torch.empty(c, dtype=torch.float64, device="cuda"),
                        arg=lambda t: t.data_ptr())
        exchange = halo.exchange
    else:
        from qgdsolver_amd.halo import HostStaged

        class Staged(HostStaged, SlabHalo):
            """host-staged variant for the gloo debugging mode"""
            torch = __import__("torch")

        halo = Staged(case, rank, world, dist, alloc=lambda c: torch.empty(c, dtype=torch.float64), arg=None)
        exchange = halo.exchange

    ov = os.environ.get("QGD_BENCH_OVERLAP", "1")
    overlap = (world > 1) and ov != "0" and (not staged or ov == "force")
    if staged and args.halo == "native":
        raise RuntimeError("--halo native needs one GPU per rank: RCCL refuses two ranks of one communicator on the same device "
                           "(\"Duplicate GPU detected\"), so the gloo debugging mode (every rank on GPU 0) cannot stand in for it")
    native = (args.halo == "native") and world > 1
    rccl_ranks = None if (world == 1 or staged) else dist.get_world_size()   # torch's communicator (staged: gloo, no RCCL at all)
    if native:
        # the library's own transport: communicator bootstrapped like an MPI host would (id from rank 0, broadcast by the launcher)
        from qgdsolver_amd.halo import NativeComm

        def bcast(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        comm = NativeComm(local_rank, rank, world, bcast=bcast)
        rccl_ranks = comm.info()["ranks"]           # ncclCommCount of the library's communicator: what RCCL itself saw
        peers = [rank - 1 if rank > 0 else -1, rank + 1 if rank < world - 1 else -1]
        exchange = lambda: comm.exchange(case, peers)  # noqa: E731

    def plain_step():
        if native:
            comm.step(case, peers, overlapped=False)
            return
        if world == 1:
            case.step_phase(3)   # one whole step, stream-ordered: the fused face + cell kernel when the case uses it (config.fused)
            return
        case.step_phase(0)   # flux assembly
        case.step_phase(1)   # cell update + boundary refresh (fixed deltaT: no global reduction needed)
        exchange()           # pack/unpack on the compute stream: strictly after the update, before the next assembly

    if overlap:
        halo_stream = torch.cuda.Stream()

        def step():
            # exchange hidden behind the bulk of the cell update (boundary layer of the shard is updated first)
            if native:
                comm.step(case, peers, overlapped=True)   # the library's own halo stream and events
            else:
                halo.step_overlapped(torch, stream, halo_stream)
    else:
        step = plain_step

    def owned_checksum():
        plane = n * n
        r = case.field("rho")[plane * (lo - k_lo): plane * (hi - k_lo)]
        t = torch.tensor([float(r.sum()), float((r * r).sum())], dtype=torch.float64, device="cpu" if ctrl_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t]

    exchange()  # ghost cells start from their owners' records
    selfcheck = None
    if overlap:
        # insurance for the overlapped exchange: 3 steps in the plain order and 3 in the overlapped order from the same
        # state must give the same owned-cell checksums; otherwise fall back to the plain order and say so
        if not native:
            case.set_halo_stream(stream.cuda_stream)
        for _ in range(3):
            plain_step()
        torch.cuda.synchronize()
        ref = owned_checksum()
        case.set_fields(*init_fields)
        exchange()
        torch.cuda.synchronize()
        if not native:
            case.set_halo_stream(halo_stream.cuda_stream)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        got = owned_checksum()
        selfcheck = max(abs(a - b) / abs(b) for a, b in zip(got, ref))
        if not selfcheck <= 1e-12:
            overlap = False
            step = plain_step
            if not native:
                case.set_halo_stream(stream.cuda_stream)
        # restart from the initial state so that runs at every N cover the same steps
        case.set_fields(*init_fields)
        torch.cuda.synchronize()
        if overlap and not native:
            halo_stream.wait_stream(stream)
            with torch.cuda.stream(halo_stream):
                exchange()
            stream.wait_stream(halo_stream)
        else:
            exchange()
    init_fields = None
    for _ in range(args.warmup):
        step()
    case.timing(False)  # the headline runs without per-launch events; kernel times come from a second pass below
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctrl_host else "cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    checksum = None
    if args.check or world > 1:   # N > 1: always, so that every point of a scaling run can be compared with N = 1
        plane = n * n
        rho_owned = case.field("rho")[plane * (lo - k_lo): plane * (hi - k_lo)]
        cs = torch.tensor([float(rho_owned.sum()), float((rho_owned ** 2).sum())], dtype=torch.float64,
                          device="cpu" if ctrl_host else "cuda")
        if world > 1:
            dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        checksum = [float(x) for x in cs]

    # second pass, not part of the headline: HIP events around every launch on the kernels' own stream
    case.timing(True)
    case.timing_reset()
    for _ in range(min(args.steps, 20)):
        step()
    torch.cuda.synchronize()
    kt = {}
    for name, k in (("point", L.K_POINT), ("bpoint", L.K_BPOINT), ("face", L.K_FACE), ("bface", L.K_BFACE), ("cell", L.K_CELL),
                    ("bc", L.K_BC)):
        ms, cnt = case.kernel_time(k)
        kt[name] = {"ms_total": ms, "launches": cnt, "ms_avg": (ms / cnt if cnt else None)}
    info = case.info()
    face_bytes = FACE_BYTES_PER_FACE * n_if + FACE_BYTES_PER_CELL * n_c + FACE_BYTES_PER_POINT * n_p
    face_ms = kt["face"]["ms_avg"]
    ft = dev.face_tiles()
    fused = case.fused_info()
    if fused["fused"]:
        fused["mean_cells_per_block"] = owned_cells / fused["blocks"]
        fb = dev.fused_blocks()
        fused.update(templates=fb["templates"], brick=list(fb["brick"]), block_list_bytes=fb["blockListBytes"], template_bytes=fb["templateBytes"],
                     block_builder_s=fb["buildSeconds"])
    # what a cut costs the blocks, and the set-up of N ranks sharing one host's cores: every rank's figures in rank 0's line
    per_rank = None
    if world > 1:
        mine = torch.tensor([float(fused["blocks"]), float(fused.get("layerBlocks", 0)), float(owned_cells), t_setup,
                             float(fused.get("facesComputed", 0)), float(fused.get("cellsStaged", 0))], dtype=torch.float64,
                            device="cpu" if ctrl_host else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"blocks": int(e[0]), "layer_blocks": int(e[1]), "owned_cells": int(e[2]),
                     "mean_cells_per_block": (float(e[2]) / float(e[0])) if float(e[0]) else None, "setup_s": float(e[3]),
                     "faces_computed_per_cell": float(e[4]) / float(e[2]), "cell_records_staged_per_cell": float(e[5]) / float(e[2])}
                    for e in (x.cpu() for x in every)]
    if fused["fused"] and kt["face"]["launches"]:
        # per STEP: on a shard the fused kernel runs as two launches (the boundary-layer blocks, then the rest)
        face_ms = kt["face"]["ms_total"] / min(args.steps, 20)
    unfused_bytes = None
    if fused["fused"]:
        # ONE launch does the work of all three rows of SURVEY 8(d) -- vertex interpolation (P), face kernel (F), cell update (C).  Its
        # ALGORITHMIC bytes are those rows' bytes WITHOUT what the rows hand each other through HBM, because that is exactly what the fusion
        # removes and never moves: F: 144 B per face in (its 40 B of net fluxes stay in LDS) + every cell record once 48 B; P: 8 weights 64 +
        # 8 labels 32 + offset 4 = 100 B per vertex (the cell records are the ones F counts; the 48 B vertex record is neither written nor read
        # back); C: face ids + signs 24 + V 8 + 5 old conserved 40 + 6 primitives written 48 = 120 B per cell (the 120 B of fluxes come out of LDS).
        # 3 x 144 + 48 + 100 + 120 = 700 B per cell-step on a hex box; the three-kernel figure (1084 B, rounds 1-5's `achieved`) stays beside
        # it as `equivalent_unfused_*`: how fast the STEP is against the reference's own data flow at 8 TB/s -- not a statement about HBM.
        unfused_bytes = face_bytes + (CELL_BYTES_PER_CELL + POINT_BYTES_PER_CELL) * n_c
        face_bytes = FUSED_BYTES_PER_FACE * n_if + FUSED_BYTES_PER_CELL * n_c + FUSED_BYTES_PER_POINT * n_p
        face_kernel_name = (f"fusedFaceCellKernel ({fused['blocks']} blocks of <= 128 cells; {fused['facesComputed']} faces computed for "
                            f"{n_if} internal faces, {fused['verticesFormed']} vertex values formed for {n_p} vertices; vertex values, faces and "
                            "cell update are ONE launch)")
        # its own compulsory bytes, from what its blocks stage: per computed face 16 B of block list + weight 8 + hQGDf 8 + kind 1; per staged
        # cell RecA 48 + label 4, for own cells and cells across a face also RecB 32 + centre 24; per own cell rhoE 8 + V 8 + hQGD 8 read,
        # 88 + 8 written, 25 B of face entries; per vertex formed 8 positions 16 + 8 weights 64 + count 1 + label 4 + coordinates 24
        # round 6: the per-face positions (12 B), the own cells' face entries (25 B per cell) and the vertices' cell positions (16 B) come out of the
        # block's TEMPLATE, which blocks that are alike share (L2-resident on a structured mesh): counted once per template, not per block
        own_bytes = (21 * fused["facesComputed"] + 52 * fused["cellsStaged"] + 56 * fused["cellsStagedFull"] + 120 * n_c +
                     93 * fused["verticesFormed"] + fb["templateBytes"] * fb["templates"])
    else:
        face_kernel_name = (f"faceFluxGvp3TileKernel<{ft['facesPerTile']}> (+ faceFluxGvp3Kernel on {ft['gatherTiles']} of {ft['tiles']} tiles; "
                            "avg_launch_ms covers both launches)") if ft["facesPerTile"] else "faceFluxGvp3Kernel"
        own_bytes = OWN_BYTES_PER_FACE * n_if + OWN_BYTES_PER_CELL * n_c + OWN_BYTES_PER_POINT * n_p
    achieved = face_bytes / (face_ms * 1e-3) / 1e9 if face_ms else None

    if rank == 0:
        total_cells = n ** 3
        value = total_cells * args.steps / elapsed / 1e6
        out = {
            "metric": "Mcell-steps/s (QGDFoam explicit step)",
            "value": value,
            "unit": "Mcell-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"QGDFoam {n}^3 = {total_cells / 1e6:.1f}M-cell uniform hex box (blockMesh numbering), "
                            "GaussVolPoint, constScPrModel1, explicit diffusion, zeroGradient patches, fixed deltaT",
                "cells": total_cells,
                "cells_per_gpu": owned_cells,
                "partition": (f"{world} k-slab(s), 1 ghost plane per cut, one send/recv pair per neighbour per step"
                              + (", exchange overlapped with the cell update" if overlap else "")) if world > 1 else "single shard",
                "transport": (("gloo, host-staged (debugging mode: every rank on GPU 0)" if staged else
                               ("RCCL inside the library (qgd_case_step_sharded; torch.distributed over gloo only for the unique id, the timing "
                                "barrier and the checksum)" if native else "RCCL through torch.distributed P2P"))
                              if world > 1 else None),
                "host_threads_per_rank": int(os.environ.get("OMP_NUM_THREADS", "0")) or None,
                "stencil": "GaussVolPoint",
                "rccl_ranks": rccl_ranks,   # ranks the communicator carrying the halo messages reports (None: one rank, or gloo staging)
                "halo_message_bytes": {"per_ghost_cell": 64, "per_ghost_patch_face": 96} if world > 1 else None,
                "fused_step": fused if fused["fused"] else False,
                "env": qgd_env(),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": face_kernel_name + " (rank 0 shard)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": None,
                "algorithmic_bytes_per_launch": face_bytes,
                "avg_launch_ms": face_ms,
                "own_layout_bytes_per_launch": own_bytes,
                "own_layout_frac": (own_bytes / (face_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if face_ms else None,
                # fused step only: SURVEY 8(d)'s three-kernel bytes (1084 B per cell-step, incl. the hand-over bytes the fused launch never moves)
                # over the same launch time -- the figure rounds 1-5 reported as `achieved` / `frac`
                "equivalent_unfused_bytes_per_launch": unfused_bytes,
                "equivalent_unfused_frac": (unfused_bytes / (face_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (unfused_bytes and face_ms) else None,
                "moved_frac": None,   # traffic / avg_launch_ms / peak: filled in below when counters of this workload exist
            },
            "other_kernels": {
                "pointInterpRecKernel": {"algorithmic_bytes_per_launch": POINT_BYTES_PER_CELL * n_c, "avg_launch_ms": kt["point"]["ms_avg"],
                                         "frac": (POINT_BYTES_PER_CELL * n_c / (kt["point"]["ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                         if kt["point"]["ms_avg"] else None},
                "cellUpdateKernel": {"algorithmic_bytes_per_launch": CELL_BYTES_PER_CELL * n_c, "avg_launch_ms": kt["cell"]["ms_avg"],
                                     "frac": (CELL_BYTES_PER_CELL * n_c / (kt["cell"]["ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                     if kt["cell"]["ms_avg"] else None},
            },
            "step_roofline_frac": STEP_BYTES_PER_CELL * owned_cells * args.steps / elapsed / (HBM_PEAK_GBS * 1e9),
            "kernels_ms_avg": {k: v["ms_avg"] for k, v in kt.items()},
            "device_bytes": case.device_bytes(),
            "setup_s": t_setup if per_rank is None else max(r["setup_s"] for r in per_rank),   # N > 1: the slowest rank's (they share the host's cores)
            "per_rank": per_rank,
            "min_rho": info["minRho"],
        }
        if checksum is not None:
            out["checksum_rho"] = checksum
        if selfcheck is not None:
            out["overlap_selfcheck_rel"] = selfcheck
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                key = f"n{n}_gpus{world}"
                if key in tr and ("fusedFaceCell" in tr[key].get("kernel", "")) == bool(fused["fused"]):   # counters of the kernel that ran
                    out["roofline"]["traffic"] = tr[key]["bytes_per_launch"]
                    out["roofline"]["traffic_source"] = tr[key].get("source")
                    # a constant read from profiles/pmc_traffic.json (counters of the builder's profiling run of this
                    # workload), not counters of THIS run
                    out["roofline"]["traffic_is_static"] = True
                    if face_ms:   # bytes the memory system really moved per launch (L2 <-> fabric) over the launch time, against the peak
                        out["roofline"]["moved_frac"] = tr[key]["bytes_per_launch"] / (face_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                    # measured HBM-side bytes / algorithmic bytes, per kernel (1.0 = nothing fetched twice)
                    out["roofline"]["kernel_traffic_ratio"] = {
                        "face": tr[key]["bytes_per_launch"] / face_bytes,
                        **{k: v / (b * n_c) for k, v, b in (("point", tr[key].get("point_bytes_per_launch"), POINT_BYTES_PER_CELL),
                                                             ("cell", tr[key].get("cell_bytes_per_launch"), CELL_BYTES_PER_CELL)) if v}}
            except Exception:
                pass
    case.close()
    dev.close()
    if native:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()   # the other ranks leave now; rank 0 may still start the second set of ranks below
    if rank == 0:
        if world > 1 and not native and not staged and not args.no_native_line and under_profiler():
            out["native_transport"] = {"skipped": "a profiler's tool library is preloaded into this process tree"}
        elif world > 1 and not native and not staged and not args.no_native_line:
            left = bench_deadline_s() - (time.perf_counter() - T_START)
            try:
                out["native_transport"] = (native_transport_line(args, world, min(420.0, left - 30.0)) if left > 90.0
                                           else {"error": "no time left before QGD_BENCH_DEADLINE_S"})
            except Exception as e:   # the second set of ranks must never cost the first line
                out["native_transport"] = {"error": repr(e)}
        elif world > 1 and staged and not args.no_native_line:
            out["native_transport"] = {"skipped": "gloo debugging mode: every rank sits on GPU 0 and RCCL refuses two ranks of one communicator on one device"}
        # the headline is known from here on: say so on stderr before any child workload starts (stdout carries ONE line, at the end)
        print(f"bench.py: headline {out['value']:.1f} {out['unit']}, {out['ms_per_step']:.3f} ms per step; secondary lines next", file=sys.stderr, flush=True)
        if world == 1 and not args.no_secondary and args.workload == "qgd":
            if under_profiler():
                out["secondary"] = {"skipped": "a profiler's tool library is preloaded into this process tree: child workloads would run under it too"}
            else:
                try:
                    out["secondary"] = secondary_lines(args)
                except Exception as e:
                    out["secondary"] = {"error": repr(e)}
        if world == 1 and not args.no_dropin:
            try:
                out["dropin_fvsc"] = dropin_fvsc_line(q, args.dropin_n, 3)
            except Exception as e:  # the second line must never cost the headline
                out["dropin_fvsc"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            ranks = args.cpu_ranks or cpu_rank_budget(args.cpu_n)
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.cpu_steps, ranks)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-rank-worker":
        _cpu_rank_worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
        sys.exit(0)
    main()
