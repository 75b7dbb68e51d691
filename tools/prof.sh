#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries are copied into profiles/ by hand.
# usage: tools/prof.sh <tag> [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-counters --no-variants "$@" > $OUT/bench_stdout.log 2>&1
ls -R $OUT | head -30
