# A/B of the fused step: the product library against a compile-time variant (ab_libs/$1), parity tests of the variant first (one gpurun call)
LIB=$PWD/ab_libs/${1:-libqgd_listlds.so}
mkdir -p gpurun_out/r06var
OUT=gpurun_out/r06var/ab_$(basename $LIB .so).txt
: > $OUT
QGD_AMD_LIB=$LIB timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_case_parity_gpu.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -3 >> $OUT
for V in "QGD_X=0" "QGD_AMD_LIB=$LIB" "QGD_X=0" "QGD_AMD_LIB=$LIB"; do
  echo "== $V" >> $OUT
  env $V timeout 400 python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f ms/step %.3f kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))
" >> $OUT
done
cat $OUT
