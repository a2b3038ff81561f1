#!/bin/bash
# round 6: the headline's kernel stats and counters (scripts/collect_profiles.sh r06: rocprofv3 --kernel-trace --stats, separate --pmc passes),
# the default bench line with its secondary lines, the secondary workloads' kernel stats, and the small-size QHD lines (2 M cells: what one
# of config 5's eight GPUs holds)
#   gpurun --timeout 3000 -- 'bash scripts/collect_r06.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
KEEP=$REPO/gpurun_out/r06_summaries
mkdir -p "$KEEP"
bash scripts/collect_profiles.sh r06 > "$KEEP/collect_profiles.log" 2>&1
cp gpurun_out/prof_r06/summaries/* "$KEEP/" 2>/dev/null
for N in 400 200; do cp gpurun_out/prof_r06/bench_stats_n$N.log "$KEEP/r06_n${N}_bench_under_rocprof.log" 2>/dev/null; done
python3 bench.py > "$KEEP/r06_bench_default.json" 2> "$KEEP/bench_default.err"
python3 bench.py --workload implicit > "$KEEP/r06_bench_implicit_n200.json" 2> "$KEEP/bench_implicit.err"
python3 bench.py --workload qhd > "$KEEP/r06_bench_qhd_n200.json" 2> "$KEEP/bench_qhd.err"
python3 bench.py --workload qhd --edge 126 > "$KEEP/r06_bench_qhd_n126.json" 2> "$KEEP/bench_qhd126.err"
python3 bench.py --workload qhd --irregular > "$KEEP/r06_bench_qhd_c5.json" 2> "$KEEP/bench_c5.err"
python3 bench.py --workload qhd --irregular --edge 126 > "$KEEP/r06_bench_qhd_c5_n126.json" 2> "$KEEP/bench_c5_126.err"
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/prof_r06
for W in implicit qhd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 "$REPO/bench.py" --workload $W --steps 20 --warmup 5 > "$KEEP/r06_${W}_n200_bench_under_rocprof.log" 2>&1
  cp $(find "$OUT/stats_$W" -name "*kernel_stats.csv" | head -1) "$KEEP/r06_${W}_n200_kernel_stats.csv"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_qhd126" -- python3 "$REPO/bench.py" --workload qhd --edge 126 --steps 20 --warmup 5 > "$KEEP/r06_qhd_n126_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_qhd126" -name "*kernel_stats.csv" | head -1) "$KEEP/r06_qhd_n126_kernel_stats.csv"
ls -la "$KEEP"
