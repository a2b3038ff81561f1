#!/bin/bash
# round 5, after the fused face + cell step became the default: the headline's kernel stats and counters again (scripts/collect_profiles.sh r05),
# the A/B against the two kernels at both sizes inside the same call, and the default bench line
#   gpurun --timeout 2700 -- 'bash scripts/collect_r05_fused.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
KEEP=$REPO/gpurun_out/r05_summaries
mkdir -p "$KEEP"
bash scripts/collect_profiles.sh r05 > "$KEEP/collect_profiles.log" 2>&1
cp gpurun_out/prof_r05/summaries/* "$KEEP/" 2>/dev/null
for N in 400 200; do
  cp gpurun_out/prof_r05/bench_stats_n$N.log "$KEEP/r05_n${N}_bench_under_rocprof.log" 2>/dev/null
  bash scripts/ab_variants.sh $N QGD_FUSED=0 QGD_FUSED=1 QGD_FUSED=0 QGD_FUSED=1 > /dev/null 2>&1
  cp gpurun_out/ab_$N.txt "$KEEP/ab_fused_n$N.txt"
done
python3 bench.py > "$KEEP/r05_bench_default.json" 2> "$KEEP/bench_default.err"
tail -3 "$KEEP/collect_profiles.log"; cat "$KEEP/ab_fused_n400.txt" "$KEEP/ab_fused_n200.txt"; head -c 600 "$KEEP/r05_bench_default.json"
