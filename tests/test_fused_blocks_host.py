"""Host logic of the fused step (CPU): the block tables of qgd_setup.cpp buildFusedBlocks checked entry by entry against the mesh tables they are
made from, on bricks, ragged boxes, a jittered mesh with triangles and polygons, and a slab shard with two cuts (tests/cpp/fused_blocks_test.cpp,
compiled with plain g++ from the library's own host sources -- no HIP, no oracle)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qgdsolver_amd", "csrc")


def test_block_tables_are_consistent_with_the_mesh(tmp_path):
    exe = str(tmp_path / "fused_blocks_test")
    cmd = ["g++", "-std=c++17", "-O2", "-fopenmp", "-I", CSRC, os.path.join(ROOT, "tests", "cpp", "fused_blocks_test.cpp")] + \
          [os.path.join(CSRC, f) for f in ("qgd_mesh.cpp", "qgd_partition.cpp", "qgd_setup.cpp")] + ["-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-4000:] + r.stderr[-2000:]
