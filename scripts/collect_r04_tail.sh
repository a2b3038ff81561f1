#!/bin/bash
# round 4, last pass: the secondary lines that collect_r04.sh does not write (full-length runs) and the implicit branch's kernel stats
#   gpurun --timeout 2400 -- 'bash scripts/collect_r04_tail.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
KEEP=$REPO/gpurun_out/r04_summaries3
mkdir -p "$KEEP"
python3 bench.py --workload implicit > "$KEEP/r04_bench_implicit_n200.json" 2> "$KEEP/bench_implicit.err"
python3 bench.py --workload qhd --irregular > "$KEEP/r04_bench_qhd_c5.json" 2> "$KEEP/bench_c5.err"
python3 bench.py --workload qhd --irregular --implicit-diffusion > "$KEEP/r04_bench_qhd_c5_implicit.json" 2> "$KEEP/bench_c5i.err"
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/prof_r04_tail
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_implicit" -- python3 "$REPO/bench.py" --workload implicit --steps 20 --warmup 5 > "$KEEP/r04_implicit_n200_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_implicit" -name "*kernel_stats.csv" | head -1) "$KEEP/r04_implicit_n200_kernel_stats.csv"
ls -la "$KEEP"
cd "$REPO"
python3 bench.py > "$KEEP/r04_bench_default.json" 2> "$KEEP/bench_default.err"
ls -la "$KEEP"
