#!/bin/bash
# tools/stepstat.py (wave-level node / leaf steps per ray kind) for every stat build under build/stat/, atrium and street
cd $GRAFT_REPO_ROOT
for so in build/stat/libfovpt_*.so; do
  n=$(basename $so .so | sed s/libfovpt_//)
  FOVPT_SO=$PWD/$so timeout -k 10 200 python tools/stepstat.py > gpurun_out/r4ss_${n}_atrium.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/r4ss_${n}_atrium.log; exit 1; }
  echo "== $n atrium"; grep "closest\|any-hit" gpurun_out/r4ss_${n}_atrium.log
  FOVPT_SCENE=street FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/stepstat.py > gpurun_out/r4ss_${n}_street.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/r4ss_${n}_street.log; exit 1; }
  echo "== $n street"; grep "closest\|any-hit" gpurun_out/r4ss_${n}_street.log
done
