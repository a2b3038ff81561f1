#!/bin/bash
# A/B of two builds of libqgd_amd.so on the QHD bench line inside ONE gpurun call (boxes of the pool differ by 3-4 %):
#   scripts/ab_qhd_libs.sh <edge> <old.so> <new.so> [extra bench flags]
edge=$1; shift; old=$1; shift; new=$1; shift
for L in "$old" "$new" "$old" "$new"; do
    echo "== $L"
    QGD_AMD_LIB=$PWD/$L timeout 600 python bench.py --workload qhd --edge "$edge" --steps 30 --warmup 5 "$@" 2>&1 | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("Mcs/s %.1f  ms/step %.3f  level-0 sweep %.4f ms (frac %.3f)  phases %s  iterations %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["phase_ms"], d["config"]["pressure_iterations_per_step"]))'
done
