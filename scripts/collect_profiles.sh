#!/bin/bash
# Regenerates the judged summaries under profiles/ on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash scripts/collect_profiles.sh r02'
# 1. rocprofv3 --kernel-trace --stats of the default bench workload (400^3) and of one 8-GPU shard's size (200^3)
# 2. separate --pmc passes (L2<->fabric read requests by size; write requests) for the HBM-side bytes per launch
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for N in 400 200; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_n$N" -- python3 "$REPO/bench.py" --edge $N --steps 50 --warmup 10 --no-cpu-baseline --no-dropin --no-secondary > "$OUT/bench_stats_n$N.log" 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc_rd_n$N" -- python3 "$REPO/bench.py" --edge $N --steps 6 --warmup 2 --no-cpu-baseline --no-dropin --no-secondary > "$OUT/bench_pmc_rd_n$N.log" 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$OUT/pmc_wr_n$N" -- python3 "$REPO/bench.py" --edge $N --steps 6 --warmup 2 --no-cpu-baseline --no-dropin --no-secondary > "$OUT/bench_pmc_wr_n$N.log" 2>&1
done
python3 "$REPO/scripts/pmc_summarise.py" "$OUT" "$TAG"
ls -la "$OUT"
