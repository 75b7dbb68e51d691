#!/bin/bash
# tools/ab_cfg.sh "<configs>": tools/configs.py <configs> under every build/libfovpt_*.so variant (stops at the first failure)
cd $GRAFT_REPO_ROOT
for so in build/libfovpt_*.so; do
  n=$(basename $so .so | sed s/libfovpt_//)
  FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/configs.py $1 > gpurun_out/abcfg_$n.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/abcfg_$n.log; exit 1; }
  echo "== $n"; grep -v amdgpu.ids gpurun_out/abcfg_$n.log
done
