#!/bin/bash
# HBM traffic counters for the bench command: two separate rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950), kernel trace only.  usage: tools/pmc.sh <tag>
set -e
TAG=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C
  mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o $C -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-counters --no-variants "$@" > $OUT/stdout.log 2>&1 || { tail -5 $OUT/stdout.log; exit 1; }
done
python3 tools/pmc_summarize.py $TAG
