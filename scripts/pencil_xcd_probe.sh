#!/bin/bash
# cell orders whose bundles match the XCD runs (done properly): rows bundled BY x BZ AND one bundle of tiles per XCD run,
# so that the j+1 / k+1 reuse of a record is a hit in the L2 of the XCD that fetched it.  Kernel times per variant; with "pmc" as first
# argument also the L2 <-> fabric requests of the three kernels (one rocprofv3 --pmc pass each).
#   usage: pencil_xcd_probe.sh [pmc] N "order xcdRun" ...      e.g.  pencil_xcd_probe.sh 400 "natural 16" "pencil:4:8 150"
PMC=0; if [ "$1" = "pmc" ]; then PMC=1; shift; fi
N=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for V in "$@"; do
  set -- $V; ORDER=$1; RUN=$2
  echo "== $ORDER QGD_XCD_RUN=$RUN"
  if [ $PMC = 1 ]; then
    QGD_XCD_RUN=$RUN bash $REPO/scripts/pmc_cmd.sh "${ORDER//:/_}_$RUN" $REPO/scripts/pmc_groups_ea.txt scripts/order_probe.py $N $ORDER 2>&1 | tail -4
  else
    (cd $REPO && QGD_XCD_RUN=$RUN timeout 600 python scripts/order_probe.py $N $ORDER 2>&1 | tail -1)
  fi
done
