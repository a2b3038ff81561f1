# QHDFoam step with the U / T equations on the cell blocks (QGD_QHD_FUSED, default 1) against the separate kernels (=0), one gpurun call
mkdir -p gpurun_out/r06qf
OUT=gpurun_out/r06qf/ab.txt
: > $OUT
for ARGS in "--edge 200" "--edge 126" "--irregular --edge 252"; do
for V in 0 1 0 1; do
  echo "== $ARGS QGD_QHD_FUSED=$V" >> $OUT
  QGD_QHD_FUSED=$V timeout 600 python bench.py --workload qhd $ARGS --steps 20 --warmup 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f ms/step %.3f phases %s iterations %s fused %s setup %.1f s' % (d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in (d['phase_ms'] or {}).items()}, d['config']['pressure_iterations_per_step'], d['config']['fused_step']['fusedAdvance'], d['setup_s']))
" >> $OUT
done
done
cat $OUT
