#!/bin/bash
# A/B of two builds of libqgd_amd.so on the implicit-branch bench lines inside ONE gpurun call:
#   scripts/ab_implicit_libs.sh <old.so> <new.so>
old=$1; new=$2
for L in "$old" "$new" "$old" "$new"; do
    echo "== $L"
    QGD_AMD_LIB=$PWD/$L timeout 600 python bench.py --workload implicit --steps 30 --warmup 5 2>/dev/null | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("implicit  Mcs/s %.1f  ms/step %.3f  kernel %.4f ms  iterations U %s e %s  stalled %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"]["iterations_U"], d["config"]["iterations_e"], d["config"].get("stalled_steps")))'
    QGD_AMD_LIB=$PWD/$L timeout 600 python bench.py --workload qhd --implicit-diffusion --steps 30 --warmup 5 2>/dev/null | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("qhd impl  Mcs/s %.1f  ms/step %.3f  implicit iterations %s  pressure %s" % (d["value"], d["ms_per_step"], d["config"]["implicit_iterations"], d["config"]["pressure_iterations_per_step"]))'
done
