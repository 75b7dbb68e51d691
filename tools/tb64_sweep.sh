#!/bin/bash
# single-wave workgroups for k_traverse (-DFOVPT_TBLOCK=64): the hardware dispatcher hands out waves as slots free up
cd $GRAFT_REPO_ROOT
export FOVPT_GRID_SHADE=8
run() { # name so grid shadowgrid
  FOVPT_SO=$PWD/build/libfovpt_$2.so FOVPT_GRID=$3 FOVPT_GRID_SHADOW=$4 timeout -k 10 150 python tools/quick_perf.py ${NTRI:-262144} 40 > gpurun_out/tb64_$1.log 2>&1 || { echo "$1 FAILED"; tail -5 gpurun_out/tb64_$1.log; exit 1; }
  echo "$1: $(grep 'profile 0' gpurun_out/tb64_$1.log | cut -d' ' -f3-5) | $(grep 'per-frame' gpurun_out/tb64_$1.log) | $(grep 'accum mean' gpurun_out/tb64_$1.log)"
}
run base_8_6 base 8 6 && run tb64_8_6 tb64 8 6 && run tb64_16_6 tb64 16 6 && run tb64_32_6 tb64 32 6 && run tb64_64_6 tb64 64 6 && run tb64_32_12 tb64 32 12 && run tb64_64_16 tb64 64 16 && run base_8_6b base 8 6
