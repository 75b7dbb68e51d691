#!/bin/bash
# per-dispatch PMC dump of one frame: tools/pmc3.sh <tag> "<counters>"
set -e
TAG=$1; CNT=$2
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc3_$TAG
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-counters --no-variants > $OUT/stdout.log 2>&1 || { tail -5 $OUT/stdout.log; exit 1; }
python3 - <<PY
import csv, re, collections
rows=list(csv.DictReader(open("$OUT/p_counter_collection.csv")))
by=collections.OrderedDict()
for r in rows:
    m=re.search(r"\b(k_[a-z_0-9]+)(?:<[^>]*>)?\s*\(", r["Kernel_Name"])
    if not m: continue
    d=by.setdefault(int(r["Dispatch_Id"]), {"k":m.group(1), "dur":(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000})
    d[r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(by)
# last full frame: from last k_generate
g=[i for i in ids if by[i]["k"]=="k_generate"][-2]
for i in ids:
    if i<g: continue
    d=by[i]
    if d["k"]=="k_generate" and i!=g: break
    extra=""
    if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d and d["SQ_ACTIVE_INST_VALU"]>0:
        extra=" lane_util=%.1f%%"%(100*d["SQ_THREAD_CYCLES_VALU"]/(64*d["SQ_ACTIVE_INST_VALU"]))
    print(d["k"], "dur_us=%.0f"%d["dur"], {k:v for k,v in d.items() if k not in ("k","dur")}, extra)
PY
