#!/bin/bash
# usage: qhd_knob_sweep.sh n "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = defaults): QHD step time and pressure iterations per setting
n=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "== $v"
  env $e timeout 300 python scripts/qhd_step_timing.py $n 4 2>&1 | tail -1 | sed -e 's/, .pInitialResidual.*mgLevels/ mgLevels/'
done
