#!/bin/bash
# tools/ab_repeat.sh <n> [tris] [frames]: every build/libfovpt_*.so variant n times, interleaved (drift cancels); stops at the first failure
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${1:-3}); do
  for so in build/libfovpt_*.so; do
    n=$(basename $so .so | sed s/libfovpt_//)
    FOVPT_SO=$PWD/$so timeout -k 10 120 python tools/quick_perf.py ${2:-262144} ${3:-200} > gpurun_out/abr_${n}_$r.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/abr_${n}_$r.log; exit 1; }
    echo "$n run $r: $(grep 'profile 0' gpurun_out/abr_${n}_$r.log | cut -d' ' -f3-4)"
  done
done
