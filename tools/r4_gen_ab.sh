#!/bin/bash
# k_generate over a rank's own tiles (product) against over all sample slots (-DFOVPT_V_GEN_OWNED=0), tile-sharded C3
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in fovpathtracing_optixcodelatest_amd/csrc/libfovpt.so build/libfovpt_genall.so; do
    echo "== $v run $r: $(FOVPT_SO=$PWD/$v timeout -k 10 200 python tools/shard_perf.py 2>&1 | grep 'rank 0' | sed 's/ rank 0: / /; s/ ms.frame.*//' | tr '\n' ';')"
  done
done
FRAMES=24 SPAN=2 bash tools/shard_timeline.sh 8 > gpurun_out/r4_shard8_gen_owned.txt 2>&1; cat gpurun_out/r4_shard8_gen_owned.txt
