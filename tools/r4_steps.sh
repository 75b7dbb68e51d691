#!/bin/bash
# step counts of a stat build ($1) under the environment given in the remaining arguments, atrium and street
cd $GRAFT_REPO_ROOT
so=$1; shift
tag=$(echo "$*" | tr ' =' '__')
env "$@" FOVPT_SO=$PWD/$so timeout -k 10 200 python tools/stepcount.py 262144 > gpurun_out/r4s_${tag}_atrium.log 2>&1 || { echo "stepcount FAILED"; tail -5 gpurun_out/r4s_${tag}_atrium.log; exit 1; }
echo "== $tag atrium"; grep "bounce\|scene\|fovpt bvh" gpurun_out/r4s_${tag}_atrium.log
env "$@" FOVPT_SCENE=street FOVPT_SO=$PWD/$so timeout -k 10 300 python tools/stepcount.py 3800000 > gpurun_out/r4s_${tag}_street.log 2>&1 || { echo "stepcount FAILED"; tail -5 gpurun_out/r4s_${tag}_street.log; exit 1; }
echo "== $tag street"; grep "bounce\|scene\|fovpt bvh" gpurun_out/r4s_${tag}_street.log
