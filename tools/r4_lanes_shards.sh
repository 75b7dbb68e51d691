#!/bin/bash
# frames in flight (FOVPT_LANES stream pairs) on the tile-sharded frames: does a 1/N shard, whose launches are each one wave's life,
# fill the GPU with more than two frames in flight?
cd $GRAFT_REPO_ROOT
for l in 2 3 4; do
  FOVPT_LANES=$l timeout -k 10 200 python tools/shard_perf.py > gpurun_out/lanes_shard_$l.txt 2>&1 || { echo "shard_perf lanes $l FAILED"; tail -5 gpurun_out/lanes_shard_$l.txt; exit 1; }
  echo "== lanes $l"; grep "world" gpurun_out/lanes_shard_$l.txt | grep "rank 0"
done
for l in 2 4; do
  FOVPT_LANES=$l timeout -k 10 400 python tools/shard_perf_big.py atrium > gpurun_out/lanes_shard_big_$l.txt 2>&1 || { echo "shard big lanes $l FAILED"; tail -5 gpurun_out/lanes_shard_big_$l.txt; exit 1; }
  echo "== lanes $l"; grep "world" gpurun_out/lanes_shard_big_$l.txt | grep "rank 0"
done
