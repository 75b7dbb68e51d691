#!/bin/bash
# more reinsertion rounds on the street: rounds cap 12 / 24 / 40 with the stop threshold at 0.15 % (product) and 0.03 %
cd $GRAFT_REPO_ROOT
P=fovpathtracing_optixcodelatest_amd/csrc/libfovpt.so
for v in "$P 12" "$P 24" "build/libfovpt_restop.so 24" "build/libfovpt_restop.so 40"; do
  set -- $v
  for sc in street atrium; do
    log=gpurun_out/re_more_$(basename $1 .so)_$2_$sc.log
    FOVPT_BVH_VERBOSE=1 FOVPT_REINSERT=$2 FOVPT_SCENE=$sc FOVPT_QP_PROFILES=0 FOVPT_SO=$PWD/$1 timeout -k 10 300 python tools/quick_perf.py 3800000 40 > $log 2>&1 || { echo "$v FAILED"; tail -5 $log; exit 1; }
    echo "$sc $(basename $1) rounds<=$2: $(grep 'profile 0' $log | cut -d' ' -f3-4) | $(grep -c 'round' $log) round lines | $(grep 'bvh nodes' $log | cut -d' ' -f1-12)"
    grep -i "round" $log | tail -2
  done
done
