#!/bin/bash
# sharded frames: one occlusion launch of a first-lane job on the second lane's shadow stream (FOVPT_SPREAD_OCCLUSION=1, product) or not (0)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 0 1; do
  echo "== spread $v run $r: $(FOVPT_SPREAD_OCCLUSION=$v timeout -k 10 200 python tools/shard_perf.py 2>&1 | grep 'rank' | sed 's/: / /; s/ ms.frame.*//' | tr '\n' ';')"
done; done
for v in 0 1; do
  echo "== spread $v: $(FOVPT_SPREAD_OCCLUSION=$v timeout -k 10 400 python tools/shard_perf_big.py atrium 2>&1 | grep world | sed 's/: / /; s/ ms.frame.*//' | tr '\n' ';')"
done
