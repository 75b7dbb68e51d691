#!/bin/bash
# the tables of DESIGN.md on the current code: all configurations, one frame sharded over N ranks, the C4 / C5 tile splits
cd $GRAFT_REPO_ROOT
T=${1:-r04c}
timeout -k 10 500 python tools/configs.py C2 C3 C4 C4S C5 C5S REF U4 PUBF PUBU > gpurun_out/${T}_configs.txt 2>&1 || { echo "configs FAILED"; tail -5 gpurun_out/${T}_configs.txt; exit 1; }
grep -v "amdgpu.ids" gpurun_out/${T}_configs.txt
timeout -k 10 200 python tools/shard_perf.py > gpurun_out/${T}_shard_perf.txt 2>&1 || { echo "shard_perf FAILED"; exit 1; }
grep world gpurun_out/${T}_shard_perf.txt
timeout -k 10 400 python tools/shard_perf_big.py atrium > gpurun_out/${T}_shard_big_atrium.txt 2>&1 || { echo "shard big atrium FAILED"; tail -5 gpurun_out/${T}_shard_big_atrium.txt; exit 1; }
grep -v "amdgpu.ids" gpurun_out/${T}_shard_big_atrium.txt | tail -12
timeout -k 10 400 python tools/shard_perf_big.py street > gpurun_out/${T}_shard_big_street.txt 2>&1 || { echo "shard big street FAILED"; tail -5 gpurun_out/${T}_shard_big_street.txt; exit 1; }
grep -v "amdgpu.ids" gpurun_out/${T}_shard_big_street.txt | tail -12
