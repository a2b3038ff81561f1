#!/bin/bash
# A/B inside one gpurun call: orders of the start-value extrapolations (QGD_QHD_PEXTRAP: pressure; QGD_IMPL_XEXTRAP: the implicit branches' solves)
for V in "QGD_QHD_PEXTRAP=3" "QGD_QHD_PEXTRAP=4" "QGD_QHD_PEXTRAP=3" "QGD_QHD_PEXTRAP=4"; do
  for W in "" "--irregular"; do
  echo "== $V qhd $W"
  env $V python bench.py --workload qhd $W --steps 20 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  iterations %s  second pass %s phase_ms %s' % (d['ms_per_step'], d['config']['pressure_iterations_per_step'], d.get('pressure_iterations_second_pass'), {k: round(v,2) for k,v in (d.get('phase_ms') or {}).items()}))
"
  done
done
for V in 3 4 3 4; do
  echo "== QGD_IMPL_XEXTRAP=$V"
  QGD_IMPL_XEXTRAP=$V python bench.py --workload implicit --steps 30 --warmup 10 2>/dev/null | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("implicit  ms/step %.3f  iterations U %s e %s  unconverged %s stalled %s" % (d["ms_per_step"], d["config"]["iterations_U"], d["config"]["iterations_e"], d["config"].get("unconverged_steps"), d["config"].get("stalled_steps")))'
  QGD_IMPL_XEXTRAP=$V python bench.py --workload qhd --implicit-diffusion --steps 30 --warmup 10 2>/dev/null | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("qhd impl  ms/step %.3f  implicit iterations %s  pressure %s unconverged %s" % (d["ms_per_step"], d["config"]["implicit_iterations"], d["config"]["pressure_iterations_per_step"], d["config"]["implicit_unconverged_steps"]))'
done
