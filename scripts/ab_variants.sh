#!/bin/bash
# A/B timing of scheduling / layout switches on one box (gpurun): bench kernels at edge N for each environment variant.
N=${1:-400}
OUT=gpurun_out/ab_$N.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" python bench.py --edge $N --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.1f ms/step %.3f  kernels %s' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms_avg'].items() if v}))
" >> $OUT
}
run QGD_DUMMY=1
run QGD_FLUX_LABEL_ORDER=1
for R in 1 8 32 128 512 2048; do run QGD_XCD_RUN=$R; done
run QGD_DUMMY=2
cat $OUT
