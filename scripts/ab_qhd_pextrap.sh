for V in ${VARIANTS:-QGD_QHD_PEXTRAP=0 QGD_QHD_PEXTRAP=1 QGD_QHD_PEXTRAP=2 QGD_QHD_PEXTRAP=1 QGD_QHD_PEXTRAP=2}; do
  for W in "" "--irregular" "--implicit-diffusion"; do
  echo "== $V $W"
  env $V python bench.py --workload qhd $W --steps 20 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  iterations %s  second pass %s phase_ms %s' % (d['ms_per_step'], d['config']['pressure_iterations_per_step'], d.get('pressure_iterations_second_pass'), {k: round(v,2) for k,v in (d.get('phase_ms') or {}).items()}))
"
  done
done
