#!/bin/bash
# frames in flight: FOVPT_LANES (main, shadow) stream pairs, with the runtime's default of 4 hardware queues and with 8
cd $GRAFT_REPO_ROOT
for q in 4 8; do
  for l in 1 2 3 4; do
    GPU_MAX_HW_QUEUES=$q FOVPT_LANES=$l timeout -k 10 150 python tools/quick_perf.py ${NTRI:-262144} 200 > gpurun_out/lanes_${q}_$l.log 2>&1 || { echo "q$q l$l FAILED"; tail -5 gpurun_out/lanes_${q}_$l.log; exit 1; }
    echo "hwq $q lanes $l: $(grep 'profile 0' gpurun_out/lanes_${q}_$l.log | cut -d' ' -f3-4) | $(grep 'accum mean' gpurun_out/lanes_${q}_$l.log)"
  done
done
