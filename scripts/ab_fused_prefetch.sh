# A/B of three latency levers of the fused step at 400^3 (one gpurun call): QGD_FU_PREFETCH (lists of a later block touched),
# QGD_FU_PREFETCH_RECORDS (its records too), QGD_FU_ROTATE (wave roles rotate with the block)
mkdir -p gpurun_out/r06pf
OUT=gpurun_out/r06pf/ab2.txt
: > $OUT
for V in "QGD_X=0" "QGD_FU_ROTATE=1" "QGD_FU_ROTATE=2" "QGD_X=0" "QGD_FU_ROTATE=3" "QGD_FU_PREFETCH=64 QGD_FU_PREFETCH_RECORDS=32" "QGD_FU_PREFETCH=32 QGD_FU_PREFETCH_RECORDS=12" "QGD_X=0" "QGD_FU_ROTATE=1"; do
  echo "== $V" >> $OUT
  env $V timeout 400 python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f ms/step %.3f kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))
" >> $OUT
done
QGD_FU_ROTATE=1 QGD_FU_PREFETCH=32 QGD_FU_PREFETCH_RECORDS=12 timeout 600 python -m pytest tests/test_fused_step_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $OUT
cat $OUT
