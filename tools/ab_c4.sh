#!/bin/bash
# tools/ab.sh for the Bistro-class C4 and stereo C5 configurations (tools/configs.py)
cd $GRAFT_REPO_ROOT
for so in build/libfovpt_*.so; do
  n=$(basename $so .so | sed s/libfovpt_//)
  echo "$n: $(FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/configs.py C4 C5 2>&1 | cut -c1-110 | tr '\n' '|')"
done
