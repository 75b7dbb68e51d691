#!/bin/bash
# tuning: occlusion-kernel blocks per CU (FOVPT_GRID_SHADOW) on the C3 frame
cd $GRAFT_REPO_ROOT
for g in 2 3 4 6 8; do
  FOVPT_GRID_SHADOW=$g timeout -k 10 120 python tools/quick_perf.py 262144 20 > gpurun_out/gs_$g.log 2>&1
  echo "grid_shadow $g: $(grep 'profile 0' gpurun_out/gs_$g.log | cut -d' ' -f3-9) | $(grep 'per-frame' gpurun_out/gs_$g.log)"
done
