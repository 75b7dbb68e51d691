#!/bin/bash
# state sets in rotation (FOVPT_SETS) with two lanes: 2 = a lane's next job waits for its previous job's resolve (rounds 3-4),
# 3 / 4 = it does not
cd $GRAFT_REPO_ROOT
for s in 2 3 4; do
  FOVPT_SETS=$s timeout -k 10 200 python tools/shard_perf.py > gpurun_out/sets_shard_$s.txt 2>&1 || { echo "shard_perf sets $s FAILED"; tail -5 gpurun_out/sets_shard_$s.txt; exit 1; }
  echo "== sets $s"; grep "world" gpurun_out/sets_shard_$s.txt | grep "rank 0"
  FOVPT_SETS=$s timeout -k 10 150 python tools/quick_perf.py 262144 200 > gpurun_out/sets_quick_$s.txt 2>&1 || { echo "quick sets $s FAILED"; tail -5 gpurun_out/sets_quick_$s.txt; exit 1; }
  grep "profile 0\|accum mean" gpurun_out/sets_quick_$s.txt
done
for s in 2 4; do
  FOVPT_SETS=$s timeout -k 10 400 python tools/shard_perf_big.py atrium > gpurun_out/sets_shard_big_$s.txt 2>&1 || { echo "shard big sets $s FAILED"; tail -5 gpurun_out/sets_shard_big_$s.txt; exit 1; }
  echo "== sets $s"; grep "world" gpurun_out/sets_shard_big_$s.txt | grep "rank 0"
done
