#!/bin/bash
# generic PMC pass: tools/pmc2.sh <tag> "<counters>"
set -e
TAG=$1; CNT=$2
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2_$TAG
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-counters --no-variants > $OUT/stdout.log 2>&1 || { tail -5 $OUT/stdout.log; exit 1; }
python3 - <<PY
import csv, re, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for row in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    m=re.search(r"\b(k_[a-z_0-9]+)\s*\(", row["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(m.group(1),row["Counter_Name"])]+=1
for k in sorted(acc):
    if k in ("k_trace","k_shadow","k_shade","k_generate","k_resolve"):
        print(k, {c: round(v/n[(k,c)],1) for c,v in acc[k].items()})
PY
