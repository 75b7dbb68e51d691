#!/bin/bash
# chains_per_frame 1 (two frames in flight) against 2 (two chains per frame, one frame in flight) over the configurations
cd $GRAFT_REPO_ROOT
for c in 1 2; do
  echo "== FOVPT_CHAINS=$c"
  FOVPT_CHAINS=$c timeout -k 10 500 python tools/configs.py C2 C3 C4 C4S C5 C5S REF U4 PUBF 2>&1 | grep -v amdgpu.ids | cut -d, -f1-4
done
