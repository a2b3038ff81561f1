#!/bin/bash
# Chebyshev smoothing of the smoothed-aggregation cycle (QGD_MG_CHEB = lambda_max / lambda_min of the smoothed interval) on the QHD bench line:
#   scripts/qhd_cheb_sweep.sh <edge> [bench flags]
edge=$1; shift
for K in "-" "QGD_MG_CHEB=3" "QGD_MG_CHEB=4" "QGD_MG_CHEB=6" "QGD_MG_CHEB=10" "QGD_MG_CHEB=4 QGD_MG_CHEB_LMAX=2.2" "QGD_MG_CHEB=6 QGD_MG_NU=3" "-"; do
    echo "== $K"
    ( [ "$K" != "-" ] && export $K; timeout 600 python bench.py --workload qhd --edge "$edge" --steps 30 --warmup 5 "$@" 2>&1 | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("Mcs/s %.1f  ms/step %.3f  solve %.3f ms  iterations %s  later %s" % (d["value"], d["ms_per_step"], d["phase_ms"]["solve"], d["config"]["pressure_iterations_per_step"], d["pressure_iterations_second_pass"]))' )
done
