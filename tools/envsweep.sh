#!/bin/bash
# tools/envsweep.sh VAR "v1 v2 ..." [tris] [frames]: tools/quick_perf.py under each value of an environment variable
# (stops at the first failure)
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2
for v in $VALS; do
  env $VAR=$v timeout -k 10 120 python tools/quick_perf.py ${3:-262144} ${4:-40} > gpurun_out/sweep_${VAR}_$v.log 2>&1 || { echo "$VAR=$v FAILED"; tail -5 gpurun_out/sweep_${VAR}_$v.log; exit 1; }
  echo "$VAR=$v: $(grep 'profile 0' gpurun_out/sweep_${VAR}_$v.log | cut -d' ' -f3-9) | $(grep 'per-frame' gpurun_out/sweep_${VAR}_$v.log)"
done
