for V in 0 3 0 3; do
  echo "== QGD_IMPL_XEXTRAP=$V"
  QGD_IMPL_XEXTRAP=$V python bench.py --workload qhd --implicit-diffusion --steps 30 --warmup 10 2>/dev/null | grep '^{' | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("qhd impl  Mcs/s %.1f  ms/step %.3f  implicit iterations %s  pressure %s unconverged %s" % (d["value"], d["ms_per_step"], d["config"]["implicit_iterations"], d["config"]["pressure_iterations_per_step"], d["config"]["implicit_unconverged_steps"]))'
done
