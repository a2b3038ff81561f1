#!/bin/bash
# round 4: everything judged under profiles/r04_* in one gpurun call; only the summaries travel back (gpurun merges at most 64 MiB)
#   gpurun --timeout 3000 -- 'bash scripts/collect_r04.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
KEEP=$REPO/gpurun_out/r04_summaries
mkdir -p "$KEEP"
bash scripts/collect_profiles.sh r04 > "$KEEP/collect_main.log" 2>&1
cp gpurun_out/prof_r04/summaries/* "$KEEP/" 2>/dev/null
cp gpurun_out/prof_r04/bench_stats_n400.log "$KEEP/r04_n400_bench_under_rocprof.log" 2>/dev/null
cp gpurun_out/prof_r04/bench_stats_n200.log "$KEEP/r04_n200_bench_under_rocprof.log" 2>/dev/null
rm -rf gpurun_out/prof_r04
bash scripts/collect_secondary_pmc.sh r04 > "$KEEP/collect_secondary.log" 2>&1
cp gpurun_out/pmc_secondary_r04/r04_pmc_secondary.json "$KEEP/" 2>/dev/null
rm -rf gpurun_out/pmc_secondary_r04
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/prof_r04_secondary
for W in implicit qhd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 "$REPO/bench.py" --workload $W --steps 20 --warmup 5 > "$KEEP/r04_${W}_n200_bench_under_rocprof.log" 2>&1
  cp $(find "$OUT/stats_$W" -name "*kernel_stats.csv" | head -1) "$KEEP/r04_${W}_n200_kernel_stats.csv"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_qhd_implicit" -- python3 "$REPO/bench.py" --workload qhd --implicit-diffusion --steps 20 --warmup 5 > "$KEEP/r04_qhd_implicit_n200_bench_under_rocprof.log" 2>&1
cp $(find "$OUT/stats_qhd_implicit" -name "*kernel_stats.csv" | head -1) "$KEEP/r04_qhd_implicit_n200_kernel_stats.csv"
cd "$REPO"
python3 bench.py > "$KEEP/r04_bench_default.json" 2> "$KEEP/bench_default.err"
ls -la "$KEEP"; du -sh gpurun_out
