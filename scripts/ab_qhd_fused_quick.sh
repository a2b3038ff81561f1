mkdir -p gpurun_out/r06qf
OUT=gpurun_out/r06qf/quick.txt
: > $OUT
timeout 600 python -m pytest tests/test_qhd_fused_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $OUT
for V in 0 1 0 1; do
  echo "== QGD_QHD_FUSED=$V" >> $OUT
  QGD_QHD_FUSED=$V timeout 600 python bench.py --workload qhd --edge 200 --steps 20 --warmup 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f ms/step %.3f phases %s fused %s' % (d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in (d['phase_ms'] or {}).items()}, d['config']['fused_step']['fusedAdvance']))
" >> $OUT
done
cat $OUT
