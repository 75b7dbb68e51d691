#!/bin/bash
# tuning: blocks per CU of the main kernels / occlusion kernels for sharded frames (per-rank time, no gather)
cd $GRAFT_REPO_ROOT
for g in ${GRIDS:-"8 6" "12 6" "16 6" "10 6" "8 8"}; do
  set -- $g
  echo "grid $1 shadow $2: $(FOVPT_GRID=$1 FOVPT_GRID_SHADOW=$2 timeout -k 10 200 python tools/shard_perf.py 2>&1 | grep 'rank 0' | awk '{printf "N=%s %s  ", $2, $5}')"
done
