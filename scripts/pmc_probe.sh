#!/bin/bash
# diagnostic counters of the three big kernels (one --pmc pass per group): what besides HBM keeps the face kernel busy?
#   gpurun --timeout 1200 -- 'bash scripts/pmc_probe.sh 200 [file with one counter group per line]'
set -u
N=${1:-200}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_probe_n$N
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
while read -r GROUP; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/g$i" -- python3 "$REPO/bench.py" --edge $N --steps 4 --warmup 2 --no-cpu-baseline --no-dropin --no-secondary > "$OUT/g$i.log" 2>&1
  echo "group $i ($GROUP): rc=$?"
done < "${2:-$REPO/scripts/pmc_groups_l1.txt}"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(s in k for s in ("faceFluxGvp3", "cellUpdateKernel", "pointInterpRec", "fusedFaceCell")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} {sum(v) / len(v):18.1f}  (n={len(v)})")
PY
