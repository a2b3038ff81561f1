#!/bin/bash
# round 4: everything judged under profiles/r04_* in one gpurun call
#   gpurun --timeout 3000 -- 'bash scripts/collect_r04.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
bash scripts/collect_profiles.sh r04 > gpurun_out/collect_r04_main.log 2>&1
bash scripts/collect_secondary_pmc.sh r04 > gpurun_out/collect_r04_secondary.log 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_r04
for W in implicit qhd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 "$REPO/bench.py" --workload $W --steps 20 --warmup 5 > "$OUT/bench_stats_$W.log" 2>&1
done
cd "$REPO"
python3 bench.py > gpurun_out/prof_r04/bench_default.json 2> gpurun_out/prof_r04/bench_default.err
ls gpurun_out/prof_r04 gpurun_out/pmc_secondary_r04
