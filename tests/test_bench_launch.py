"""bench.py --gpus N without a launcher around it: the parent starts N rank processes itself before it
touches torch or HIP, relays rank 0's line and the worst status.

CPU: the launcher's failure path -- children that cannot run (no HIP device here) must take the whole job down with a
non-zero status, promptly, leaving no process behind.  GPU: two ranks over gloo on the one GPU of the box (host-staged halo
messages) reproduce the one-rank run's owned-cell checksum to 1e-12."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, timeout):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    pr = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return pr, time.time() - t0


def test_launcher_fails_loudly_when_the_ranks_cannot_run():
    import qgdsolver_amd as q
    if q.device_count() > 0:
        pytest.skip("a HIP device is visible: the ranks would run")
    pr, dt = run_bench(["--gpus", "2", "--backend", "gloo", "--edge", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], 300)
    assert pr.returncode not in (0, 2), (pr.returncode, pr.stderr[-400:])   # 2 was "launch me with torch.distributed.run"
    assert not [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]


def test_launcher_and_flag_must_agree():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert pr.returncode == 2 and "disagree" in pr.stderr


def test_the_implicit_line_is_a_one_gpu_line():
    pr, _ = run_bench(["--workload", "implicit", "--gpus", "2", "--steps", "1"], 120)
    assert pr.returncode == 2 and "one-GPU line" in pr.stderr


def test_deadline_kills_hung_ranks():
    """a hung collective must not burn the caller's timeout: when QGD_BENCH_DEADLINE_S passes the relay kills every rank and reports
    status 124; a rank that dies takes the others with it at once"""
    import bench
    hang = [sys.executable, "-c", "import time; time.sleep(120)"]
    t0 = time.time()
    res = bench.launch_ranks(2, [], 2.0, cmd=hang)
    assert res["timed_out"] and res["rc"] == 124 and res["line"] is None and time.time() - t0 < 30
    # rank 1 dies after a second, rank 0 would sleep for two minutes
    die = [sys.executable, "-c", "import os, sys, time; time.sleep(1 if os.environ['RANK'] == '1' else 120); sys.exit(7)"]
    t0 = time.time()
    res = bench.launch_ranks(2, [], 60.0, cmd=die)
    assert not res["timed_out"] and res["rc"] != 0 and time.time() - t0 < 30
    # the ordinary case: rank 0's JSON line is relayed, other stdout is not a line, the launcher's own variables reach the ranks and a
    # surrounding torchrun's do not
    ok = [sys.executable, "-c", "import os, json; r = os.environ['RANK']; print('noise'); "
                                "print(json.dumps({'rank': r, 'world': os.environ['WORLD_SIZE'], 'agent': os.environ.get('TORCHELASTIC_USE_AGENT_STORE')}))"]
    os.environ["TORCHELASTIC_USE_AGENT_STORE"] = "True"
    try:
        res = bench.launch_ranks(3, [], 60.0, cmd=ok)
    finally:
        del os.environ["TORCHELASTIC_USE_AGENT_STORE"]
    assert res["rc"] == 0 and json.loads(res["line"]) == {"rank": "0", "world": "3", "agent": None}


def test_self_launch_enforces_the_deadline_from_the_environment():
    """python bench.py --gpus 2 with QGD_BENCH_DEADLINE_S: the relay parent itself (no GPU needed: the ranks are replaced by sleepers
    through the same launch_ranks the relay uses)"""
    code = ("import sys, bench; bench.launch_ranks.__defaults__ = ([sys.executable, '-c', 'import time; time.sleep(120)'],); "
            "sys.argv = ['bench.py', '--gpus', '2']; bench.self_launch(2)")
    env = dict(os.environ, QGD_BENCH_DEADLINE_S="2", PYTHONPATH=ROOT)
    t0 = time.time()
    pr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert pr.returncode == 124 and "QGD_BENCH_DEADLINE_S" in pr.stderr and time.time() - t0 < 60, (pr.returncode, pr.stderr[-300:])
    assert not [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]


def test_native_transport_is_attached_from_a_second_set_of_ranks(monkeypatch):
    """the merge logic of the second line (the real thing needs N > 1 GPUs: RCCL refuses two ranks on one device): what the fresh
    ranks print is cut to the keys the first line is compared on, a failure becomes {"error": ...} and never an exception"""
    import bench
    seen = {}

    def fake(n, argv, deadline, cmd=None):
        seen["argv"], seen["n"] = list(argv), n
        return {"line": json.dumps({"ms_per_step": 2.5, "value": 25600.0, "checksum_rho": [1.0, 2.0],
                                     "config": {"transport": "RCCL inside the library (qgd_case_step_sharded; ...)", "rccl_ranks": 8, "partition": "8 k-slab(s)"}}),
                "rc": 0, "timed_out": False, "stderr": []}

    monkeypatch.setattr(bench, "launch_ranks", fake)
    args = type("A", (), {"steps": 100, "warmup": 20, "n": 400, "backend": "nccl"})()
    nt = bench.native_transport_line(args, 8, 300.0)
    assert nt["ms_per_step"] == 2.5 and nt["checksum_rho"] == [1.0, 2.0] and nt["rccl_ranks"] == 8 and nt["transport"].startswith("RCCL inside the library")
    a = seen["argv"]
    assert seen["n"] == 8 and a[a.index("--halo") + 1] == "native" and "--no-native-line" in a and "--check" in a
    assert a[a.index("--steps") + 1] == "100" and a[a.index("--warmup") + 1] == "20" and a[a.index("--edge") + 1] == "400"
    monkeypatch.setattr(bench, "launch_ranks", lambda *a, **k: {"line": None, "rc": 124, "timed_out": True, "stderr": ["stuck"]})
    nt = bench.native_transport_line(args, 8, 5.0)
    assert "error" in nt and "5 s" in nt["error"] and nt["stderr"] == ["stuck"]


def test_env_switches_are_echoed(monkeypatch):
    import bench
    monkeypatch.setenv("QGD_XCD_RUN", "16")
    monkeypatch.setenv("NOT_OURS", "1")
    e = bench.qgd_env()
    assert e["QGD_XCD_RUN"] == "16" and "NOT_OURS" not in e


@pytest.mark.gpu
def test_two_ranks_self_launched_match_one_rank():
    common = ["--backend", "gloo", "--edge", "48", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-dropin", "--no-secondary", "--check"]
    one, _ = run_bench(["--gpus", "1"] + common, 900)
    assert one.returncode == 0, one.stderr[-800:]
    two, _ = run_bench(["--gpus", "2"] + common, 900)
    assert two.returncode == 0, two.stderr[-800:]
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["config"]["transport"].startswith("gloo") and b["config"]["cells_per_gpu"] == 48 * 48 * 24
    for x, y in zip(a["checksum_rho"], b["checksum_rho"]):
        assert abs(x - y) <= 1e-12 * abs(x), (a["checksum_rho"], b["checksum_rho"])
    assert b["roofline"]["frac"] is not None and b["value"] > 0
    # new keys: gloo staging has no RCCL communicator, the C-ABI transport cannot stand on one GPU, and says so
    assert b["config"]["rccl_ranks"] is None and "skipped" in b["native_transport"] and "one device" in b["native_transport"]["skipped"]
    assert b["config"]["halo_message_bytes"] == {"per_ghost_cell": 64, "per_ghost_patch_face": 96} and isinstance(b["config"]["env"], dict)


@pytest.mark.gpu
def test_native_halo_refuses_the_one_gpu_stand_in():
    pr, _ = run_bench(["--gpus", "2", "--backend", "gloo", "--halo", "native", "--edge", "16", "--steps", "1", "--warmup", "0",
                       "--no-cpu-baseline", "--no-dropin", "--no-secondary"], 600)
    assert pr.returncode != 0 and not [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_default_line_carries_the_secondary_workloads_and_the_env():
    """python bench.py --gpus 1 (the driver's command) appends `secondary` {qhd, implicit} from child processes, 20 timed steps each,
    and echoes every QGD_* variable; shrunk here through QGD_BENCH_SECONDARY_N, which the echo then shows"""
    env = dict(os.environ, QGD_BENCH_SECONDARY_N="24", QGD_BENCH_C5_N="20")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, BENCH, "--edge", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-dropin"], env=env,
                        capture_output=True, text=True, timeout=1200)
    assert pr.returncode == 0, pr.stderr[-800:]
    d = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["env"]["QGD_BENCH_SECONDARY_N"] == "24" and d["config"]["rccl_ranks"] is None and "native_transport" not in d
    sec = d["secondary"]
    assert set(sec) == {"qhd_n24", "qhd_implicit_n24", "implicit_n24", "adjust_n24", "qhd_c5_n20"}    # BASELINE config 5's mesh recipe, shrunk
    assert sec["adjust_n24"]["steps"] == 50 and sec["adjust_n24"]["value"] > 0 and 0 < sec["adjust_n24"]["config"]["CoNum"] <= 0.12   # Courant-number control
    assert "config-5 stand-in mesh" in sec["qhd_c5_n20"]["config"]["workload"] and sec["qhd_c5_n20"]["steps"] == 20 and sec["qhd_c5_n20"]["value"] > 0
    assert "headline" in pr.stderr                         # the headline is announced before the child workloads start
    assert sec["qhd_implicit_n24"]["config"]["implicit_iterations"]["T"] > 0 and sec["qhd_n24"]["config"]["implicit_iterations"] is None
    for key, it_key in (("qhd_n24", "pressure_iterations_per_step"), ("qhd_implicit_n24", "pressure_iterations_per_step"), ("implicit_n24", "iterations_U")):
        x = sec[key]
        assert "error" not in x, x
        assert x["steps"] == 20 and x["ms_per_step"] > 0 and x["value"] > 0 and x["config"][it_key] > 0 and x["roofline"]["avg_launch_ms"] > 0


@pytest.mark.gpu
def test_qhd_workload_on_two_self_launched_ranks():
    """python bench.py --workload qhd --gpus 2: config 5's path (sharded QHDFoam case, pressure solve with the multigrid hierarchy that
    spans the ranks) as two processes on the one GPU of the box, gloo with host-staged messages; same pressure iterations as one rank"""
    common = ["--workload", "qhd", "--edge", "32", "--steps", "2", "--warmup", "1"]   # (secondary lines belong to the qgd workload only)
    one, _ = run_bench(["--gpus", "1"] + common, 900)
    assert one.returncode == 0, one.stderr[-800:]
    two, _ = run_bench(["--gpus", "2", "--backend", "gloo"] + common, 900)
    assert two.returncode == 0, two.stderr[-1500:]
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["config"]["cells"] == 32 ** 3 and b["config"]["cells_per_gpu"] == 32 ** 3 // 2
    assert b["config"]["transport"].startswith("gloo") and b["value"] > 0
    assert 0 < b["config"]["pressure_iterations_per_step"] <= a["config"]["pressure_iterations_per_step"] + 2, (a["config"], b["config"])
    assert b["config"]["multigrid_levels"] == a["config"]["multigrid_levels"] and b["pressure_final_residual"] < 1e-8


@pytest.mark.gpu
def test_implicit_workload_line():
    """the reference's default branch as a bench line: metric, iterations of both systems, the timed matrix product"""
    pr, _ = run_bench(["--workload", "implicit", "--edge", "24", "--steps", "3", "--warmup", "1"], 600)
    assert pr.returncode == 0, pr.stderr[-800:]
    d = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["metric"].startswith("Mcell-steps/s") and d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["cells"] == 24 ** 3 and d["config"]["iterations_U"] > 0 and d["config"]["iterations_e"] > 0
    assert d["config"]["unconverged_steps"] == 0 and d["min_rho"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_launch"] == 216 * 24 ** 3 and r["traffic"] is None and "iChebKernel" in r["kernel"]
