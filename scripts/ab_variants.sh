#!/bin/bash
# A/B timing of scheduling / layout switches on one box (gpurun): bench kernels at edge N for each environment variant.
# usage: ab_variants.sh N "VAR=val VAR2=val" "VAR=val" ...   (each argument one variant; "-" = defaults)
N=${1:-400}; shift
OUT=gpurun_out/ab_$N.txt
: > $OUT
for V in "$@"; do
  echo "== $V" >> $OUT
  if [ "$V" = "-" ]; then V="QGD_DUMMY=1"; fi
  env $V python bench.py --edge $N --steps 30 --warmup 5 --no-cpu-baseline --no-dropin --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.1f ms/step %.3f  kernels %s' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms_avg'].items() if v}))
" >> $OUT
done
cat $OUT
