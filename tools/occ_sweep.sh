#!/bin/bash
# resident waves per CU of the closest-hit launches: FOVPT_GRID blocks of 256 per CU, all resident (grid-stride over the queue)
cd $GRAFT_REPO_ROOT
export FOVPT_GRID_SHADE=8
for g in 2 3 4 5 6 7 8; do
  FOVPT_SO=$PWD/build/libfovpt_base.so FOVPT_GRID=$g timeout -k 10 150 python tools/quick_perf.py ${NTRI:-262144} 40 > gpurun_out/occ_$g.log 2>&1 || { echo "$g FAILED"; tail -5 gpurun_out/occ_$g.log; exit 1; }
  echo "grid $g: $(grep 'profile 0' gpurun_out/occ_$g.log | cut -d' ' -f3-5) | $(grep 'per-frame' gpurun_out/occ_$g.log)"
done
