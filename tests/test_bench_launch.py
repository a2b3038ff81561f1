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


@pytest.mark.gpu
def test_two_ranks_self_launched_match_one_rank():
    common = ["--backend", "gloo", "--edge", "48", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-dropin", "--check"]
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


@pytest.mark.gpu
def test_qhd_workload_on_two_self_launched_ranks():
    """python bench.py --workload qhd --gpus 2: config 5's path (sharded QHDFoam case, pressure solve with the multigrid hierarchy that
    spans the ranks) as two processes on the one GPU of the box, gloo with host-staged messages; same pressure iterations as one rank"""
    common = ["--workload", "qhd", "--edge", "32", "--steps", "2", "--warmup", "1"]
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
    assert r["bound"] == "hbm" and r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_launch"] == 144 * 24 ** 3 and r["traffic"] is None
