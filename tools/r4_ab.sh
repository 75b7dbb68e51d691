#!/bin/bash
# round 4 A/B: every build/libfovpt_*.so on C3 (atrium 262 k) and on the 3.8 M-triangle street, interleaved; then the step counts of
# the stat builds under build/stat/.  Stops at the first failure (no GPU step after a failed one).
cd $GRAFT_REPO_ROOT
R=${1:-2}
for r in $(seq 1 $R); do
  for so in build/libfovpt_*.so; do
    n=$(basename $so .so | sed s/libfovpt_//)
    FOVPT_SO=$PWD/$so timeout -k 10 150 python tools/quick_perf.py 262144 200 > gpurun_out/r4_${n}_c3_$r.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/r4_${n}_c3_$r.log; exit 1; }
    echo "C3 $n run $r: $(grep 'profile 0' gpurun_out/r4_${n}_c3_$r.log | cut -d' ' -f3-4) | $(grep 'per-frame' gpurun_out/r4_${n}_c3_$r.log)"
  done
done
for r in $(seq 1 $R); do
  for so in build/libfovpt_*.so; do
    n=$(basename $so .so | sed s/libfovpt_//)
    FOVPT_SCENE=street FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/quick_perf.py 3800000 40 > gpurun_out/r4_${n}_street_$r.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/r4_${n}_street_$r.log; exit 1; }
    echo "street $n run $r: $(grep 'profile 0' gpurun_out/r4_${n}_street_$r.log | cut -d' ' -f3-4) | $(grep 'per-frame' gpurun_out/r4_${n}_street_$r.log)"
  done
done
for so in build/stat/libfovpt_*.so; do
  n=$(basename $so .so | sed s/libfovpt_//)
  FOVPT_SO=$PWD/$so timeout -k 10 200 python tools/stepcount.py 262144 > gpurun_out/r4_${n}_steps_c3.log 2>&1 || { echo "$n stepcount FAILED"; tail -5 gpurun_out/r4_${n}_steps_c3.log; exit 1; }
  echo "== $n atrium"; grep "bounce\|scene" gpurun_out/r4_${n}_steps_c3.log
  FOVPT_SCENE=street FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/stepcount.py 3800000 > gpurun_out/r4_${n}_steps_street.log 2>&1 || { echo "$n stepcount FAILED"; tail -5 gpurun_out/r4_${n}_steps_street.log; exit 1; }
  echo "== $n street"; grep "bounce\|scene" gpurun_out/r4_${n}_steps_street.log
done
