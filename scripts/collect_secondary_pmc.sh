#!/bin/bash
# HBM-side bytes per launch of the kernels in the `roofline` objects of `bench.py --workload qhd` and `--workload implicit`
# (separate --pmc passes for the read and the write requests, as scripts/collect_profiles.sh does for the headline kernels):
#   gpurun --timeout 1200 -- 'bash scripts/collect_secondary_pmc.sh r03'
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_secondary_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for W in qhd implicit qhd_c5; do
  N=200; [ "$W" = qhd_c5 ] && N=252
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/rd_$W" -- python3 "$REPO/scripts/secondary_kernel_probe.py" $W $N > "$OUT/rd_$W.log" 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$OUT/wr_$W" -- python3 "$REPO/scripts/secondary_kernel_probe.py" $W $N > "$OUT/wr_$W.log" 2>&1
  grep -h PROBE "$OUT/rd_$W.log" "$OUT/wr_$W.log"
done
python3 "$REPO/scripts/pmc_secondary_summarise.py" "$OUT" "$TAG"
