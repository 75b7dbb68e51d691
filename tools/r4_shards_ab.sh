#!/bin/bash
# the shard measures of round 4 one at a time: state sets in rotation (FOVPT_SETS), thin rounds (a filed experiment,
# tools/experiments/r04_thin_rounds.patch), with two frames in flight and with one:
# tile-sharded C3 (tools/shard_perf.py), the whole C3 frame (tools/quick_perf.py), C4 / C5 shards (tools/shard_perf_big.py atrium)
cd $GRAFT_REPO_ROOT
P=fovpathtracing_optixcodelatest_amd/csrc/libfovpt.so
V="old|build/libfovpt_cur.so|FOVPT_SETS=2
sets2|$P|FOVPT_SETS=2
sets4|$P|FOVPT_SETS=4
sets2_1lane|$P|FOVPT_SETS=2 FOVPT_LANES=1
thin_1lane|build/libfovpt_thin.so|FOVPT_SETS=2 FOVPT_LANES=1"
for r in 1 2; do
  while IFS='|' read -r n so envs; do
    env $envs FOVPT_SO=$PWD/$so timeout -k 10 200 python tools/shard_perf.py > gpurun_out/sab_${n}_$r.txt 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/sab_${n}_$r.txt; exit 1; }
    echo "== $n run $r: $(grep world gpurun_out/sab_${n}_$r.txt | grep 'rank 0' | sed 's/ rank 0: / /; s/ ms.frame.*//' | tr '\n' ';')"
    env $envs FOVPT_QP_PROFILES=0,2 FOVPT_SO=$PWD/$so timeout -k 10 150 python tools/quick_perf.py 262144 200 > gpurun_out/sabq_${n}_$r.txt 2>&1 || { echo "$n quick FAILED"; exit 1; }
    echo "   $(grep 'profile 0' gpurun_out/sabq_${n}_$r.txt | cut -d' ' -f3-4) | $(grep 'per-frame' gpurun_out/sabq_${n}_$r.txt)"
  done <<< "$V"
done
while IFS='|' read -r n so envs; do
  env $envs FOVPT_SO=$PWD/$so timeout -k 10 400 python tools/shard_perf_big.py atrium > gpurun_out/sabb_${n}.txt 2>&1 || { echo "$n big FAILED"; tail -5 gpurun_out/sabb_${n}.txt; exit 1; }
  echo "== $n: $(grep world gpurun_out/sabb_${n}.txt | grep 'rank 0' | sed 's/ rank 0: / /; s/ ms.frame.*//' | tr '\n' ';')"
done <<< "$V"
