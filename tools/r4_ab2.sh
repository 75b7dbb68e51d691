#!/bin/bash
# round 4 A/B, second form: variants are "name|library|ENV=VAL ENV=VAL" lines in the file given as $1; every variant on C3 (atrium
# 262 k) and on the 3.8 M-triangle street, interleaved, $2 repeats.  Stops at the first failure.
cd $GRAFT_REPO_ROOT
LIST=$1; R=${2:-2}
run() { # scene tag, tris, frames
  for r in $(seq 1 $R); do
    while IFS='|' read -r n so envs; do
      [ -z "$n" ] && continue
      log=gpurun_out/r4b_${n}_$1_$r.log
      env $envs FOVPT_SCENE=$1 FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/quick_perf.py $2 $3 > $log 2>&1 || { echo "$n FAILED"; tail -5 $log; exit 1; }
      echo "$1 $n run $r: $(grep 'profile 0' $log | cut -d' ' -f3-4) | $(grep 'per-frame' $log) | $(grep 'bvh nodes' $log | cut -d' ' -f1-8)"
    done < $LIST
  done
}
run atrium 262144 200 || exit 1
run street 3800000 40 || exit 1
