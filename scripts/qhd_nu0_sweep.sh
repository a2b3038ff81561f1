#!/bin/bash
# A/B inside one gpurun call: sweeps on multigrid level 0 (QGD_MG_NU0) against the coarse levels' (QGD_MG_NU), damping; QHD bench line at 200^3
# usage: qhd_nu0_sweep.sh [--irregular]
EXTRA="$@"
for V in "-" "QGD_MG_NU0=1" "QGD_MG_NU0=1 QGD_MG_NU=3" "QGD_MG_NU0=1 QGD_MG_NU=4" "QGD_MG_OMEGA=0.9" "QGD_MG_NU0=1 QGD_MG_OMEGA=0.9" "QGD_MG_NU0=1 QGD_MG_NU=3 QGD_MG_OMEGA=0.9" "-"; do
  echo "== $V"
  if [ "$V" = "-" ]; then V="QGD_DUMMY=1"; fi
  env $V python bench.py --workload qhd $EXTRA --steps 20 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  iterations %s  phase_ms %s' % (d['ms_per_step'], d['config']['pressure_iterations_per_step'], {k: round(v,2) for k,v in (d.get('phase_ms') or {}).items()}))
"
done
