#!/bin/bash
# A/B inside one gpurun call: the QHD bench line (200^3 unless --irregular / --edge) for each environment variant ("-" = defaults)
# usage: qhd_ab.sh "VAR=val VAR2=val" "VAR=val" ... [-- bench arguments]
VARS=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; VARS+=("$1"); shift; done
for V in "${VARS[@]}"; do
  echo "== $V"
  if [ "$V" = "-" ]; then V="QGD_DUMMY=1"; fi
  env $V python bench.py --workload qhd "${EXTRA[@]}" --steps 20 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  iterations %s  phase_ms %s' % (d['ms_per_step'], d['config']['pressure_iterations_per_step'], {k: round(v,2) for k,v in (d.get('phase_ms') or {}).items()}))
"
done
