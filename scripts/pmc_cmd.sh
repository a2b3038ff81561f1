#!/bin/bash
# per-kernel averages of hardware counters for an arbitrary python script: one rocprofv3 --pmc pass per line of GROUPFILE
#   gpurun -- 'bash scripts/pmc_cmd.sh TAG $GRAFT_REPO_ROOT/scripts/pmc_groups_ea.txt scripts/order_probe.py 200 natural'
set -u
TAG=$1; GROUPS_FILE=$2; shift 2
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
i=0
while read -r GROUP; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/g$i" -- python3 "$SCRIPT" "$@" > "$OUT/g$i.log" 2>&1
  echo "group $i ($GROUP): rc=$? $(grep -E 'step .* ms' "$OUT/g$i.log" | tail -1)"
done < "$GROUPS_FILE"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(s in k for s in ("faceFluxGvp3", "cellUpdateKernel", "pointInterpRec")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, "  ".join(f"{c}={sum(v) / len(v):.4g} (n={len(v)})" for c, v in sorted(d.items())))
PY
