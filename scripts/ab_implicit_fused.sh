#!/bin/bash
# A/B of the implicitDiffusion branch's block-fused assembly (QGD_IMPL_FUSED) on one box: bench.py --workload implicit at edge N
N=${1:-200}
OUT=gpurun_out/ab_implicit_fused_$N.txt
: > $OUT
for V in QGD_IMPL_FUSED=0 QGD_IMPL_FUSED=1 QGD_IMPL_FUSED=0 QGD_IMPL_FUSED=1; do
  echo "== $V" >> $OUT
  env $V python bench.py --workload implicit --edge $N --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.1f ms/step %.3f  fused %s  iterations U %s e %s' % (d['value'], d['ms_per_step'], d['config'].get('fused_assembly_of_the_U_systems'), d['config']['iterations_U'], d['config']['iterations_e']))
" >> $OUT
done
cat $OUT
