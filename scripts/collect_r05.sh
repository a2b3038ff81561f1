#!/bin/bash
# round 5: the bench lines and the secondary workloads' kernel stats (the headline's stats and counters: scripts/collect_profiles.sh r05;
# the secondary kernels' counters: scripts/collect_secondary_pmc.sh r05)
#   gpurun --timeout 2700 -- 'bash scripts/collect_r05.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
KEEP=$REPO/gpurun_out/r05_summaries
mkdir -p "$KEEP"
python3 bench.py > "$KEEP/r05_bench_default.json" 2> "$KEEP/bench_default.err"
python3 bench.py --workload implicit > "$KEEP/r05_bench_implicit_n200.json" 2> "$KEEP/bench_implicit.err"
python3 bench.py --workload qhd > "$KEEP/r05_bench_qhd_n200.json" 2> "$KEEP/bench_qhd.err"
python3 bench.py --workload qhd --irregular > "$KEEP/r05_bench_qhd_c5.json" 2> "$KEEP/bench_c5.err"
python3 bench.py --workload qhd --irregular --implicit-diffusion > "$KEEP/r05_bench_qhd_c5_implicit.json" 2> "$KEEP/bench_c5i.err"
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/prof_r05
for W in implicit qhd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 "$REPO/bench.py" --workload $W --steps 20 --warmup 5 > "$KEEP/r05_${W}_n200_bench_under_rocprof.log" 2>&1
  cp $(find "$OUT/stats_$W" -name "*kernel_stats.csv" | head -1) "$KEEP/r05_${W}_n200_kernel_stats.csv"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_qhd_implicit" -- python3 "$REPO/bench.py" --workload qhd --implicit-diffusion --steps 20 --warmup 5 > "$KEEP/r05_qhd_implicit_n200_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_qhd_implicit" -name "*kernel_stats.csv" | head -1) "$KEEP/r05_qhd_implicit_n200_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_qhd_c5" -- python3 "$REPO/bench.py" --workload qhd --irregular --steps 20 --warmup 5 > "$KEEP/r05_qhd_c5_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_qhd_c5" -name "*kernel_stats.csv" | head -1) "$KEEP/r05_qhd_c5_kernel_stats.csv"
ls -la "$KEEP"
