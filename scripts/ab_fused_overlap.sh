# QGD_FU_OVERLAP (the fused step's patch kernels on a stream of their own beside the interior blocks) on / off: parity tests, bit-identity at 128^3,
# then the default step at 400^3 and 200^3, arms alternating (one gpurun call)
mkdir -p gpurun_out/r06ov
OUT=gpurun_out/r06ov/ab.txt
: > $OUT
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_case_parity_gpu.py tests/test_golden.py tests/test_partition_gpu.py tests/test_symmetry_patches.py -q -m gpu -x 2>&1 | tail -3 >> $OUT
timeout 600 python scripts/fused_launch_shape_check.py 128 7 QGD_FU_OVERLAP 1 2>&1 | tail -2 >> $OUT
for N in 400 200; do
for V in 0 1 0 1; do
  echo "== n $N QGD_FU_OVERLAP=$V" >> $OUT
  QGD_FU_OVERLAP=$V timeout 400 python bench.py --edge $N --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f ms/step %.3f kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))
" >> $OUT
done
done
cat $OUT
