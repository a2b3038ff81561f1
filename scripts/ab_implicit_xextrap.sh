#!/bin/bash
# A/B inside one gpurun call: the start values of the implicit branch's solves (QGD_IMPL_XEXTRAP = 0..3) on the implicit bench line
for V in 0 3 2 1 0 3; do
  echo "== QGD_IMPL_XEXTRAP=$V"
  QGD_IMPL_XEXTRAP=$V python bench.py --workload implicit --steps 30 --warmup 10 2>/dev/null | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("implicit  Mcs/s %.1f  ms/step %.3f  iterations U %s e %s  unconverged %s stalled %s" % (d["value"], d["ms_per_step"], d["config"]["iterations_U"], d["config"]["iterations_e"], d["config"].get("unconverged_steps"), d["config"].get("stalled_steps")))'
done
