#!/bin/bash
# round 4, after the QHD changes (staged face passes, fused cycle hand-over): the summaries that depend on them, in one gpurun call
#   gpurun --timeout 2400 -- 'bash scripts/collect_r04_qhd.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
KEEP=$REPO/gpurun_out/r04_summaries2
mkdir -p "$KEEP"
python3 bench.py > "$KEEP/r04_bench_default.json" 2> "$KEEP/bench_default.err"
python3 bench.py --workload qhd --irregular --steps 20 --warmup 5 > "$KEEP/r04_bench_qhd_c5.json" 2> "$KEEP/bench_c5.err"
python3 bench.py --workload qhd --irregular --implicit-diffusion --steps 20 --warmup 5 > "$KEEP/r04_bench_qhd_c5_implicit.json" 2> "$KEEP/bench_c5i.err"
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/prof_r04_qhd
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_qhd" -- python3 "$REPO/bench.py" --workload qhd --steps 20 --warmup 5 > "$KEEP/r04_qhd_n200_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_qhd" -name "*kernel_stats.csv" | head -1) "$KEEP/r04_qhd_n200_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_qhd_implicit" -- python3 "$REPO/bench.py" --workload qhd --implicit-diffusion --steps 20 --warmup 5 > "$KEEP/r04_qhd_implicit_n200_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_qhd_implicit" -name "*kernel_stats.csv" | head -1) "$KEEP/r04_qhd_implicit_n200_kernel_stats.csv"
ls -la "$KEEP"
