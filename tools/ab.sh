#!/bin/bash
# run tools/quick_perf.py against every build/libfovpt_*.so variant (on the GPU box); stops at the first failure
# (after a GPU step fails no further GPU step is started)
cd $GRAFT_REPO_ROOT
for so in build/libfovpt_*.so; do
  n=$(basename $so .so | sed s/libfovpt_//)
  FOVPT_SO=$PWD/$so timeout -k 10 120 python tools/quick_perf.py ${1:-262144} ${2:-20} > gpurun_out/ab_$n.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/ab_$n.log; exit 1; }
  echo "$n: $(grep 'profile 0' gpurun_out/ab_$n.log | cut -d' ' -f3-9) | $(grep 'per-frame' gpurun_out/ab_$n.log)"
done
