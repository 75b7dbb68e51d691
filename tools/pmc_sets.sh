#!/bin/bash
# tools/pmc_sets.sh <tag> [set ...]: per-dispatch counters of one C3 frame, ONE rocprofv3 pass per counter set, each pass under
# its own timeout.  Sets are kept within the per-block slots of gfx950 (MI355X_MICROARCH.md: SQ 8, TCC 4, GRBM 2; the TCP / TA
# sets below are ones that have been collected on this stack): a request over a block's slots makes the profiler's tool library
# abort inside the profiled process's first HIP call ("Could not construct profile cfg ... error code 38: Request exceeds the
# capabilities of the hardware to collect") and the process then sits there until something kills it -- the "hang" of round 2
# (gpurun_out/pmc3_r02p1/stdout.log).  Stops at the first pass that fails.
set -e
TAG=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
declare -A SETS
SETS[sq_insts]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
SETS[sq_wait]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
SETS[sq_cycles]="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES"
SETS[sq_level]="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS"
SETS[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE"
# (four TA counters in one pass are over the TA block's slots: error code 38, measured in round 3 -- two per pass)
SETS[ta]="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
SETS[ta2]="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
SETS[tcc]="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
NAMES="$@"
[ -z "$NAMES" ] && NAMES="sq_insts sq_wait sq_cycles sq_level tcp ta ta2 tcc"
for S in $NAMES; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcset_${TAG}_$S
  mkdir -p $OUT
  timeout -k 10 150 rocprofv3 --pmc ${SETS[$S]} --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-counters --no-variants > $OUT/stdout.log 2>&1 \
    || { echo "set $S FAILED"; grep -m2 "exceeds the capabilities\|rror" $OUT/stdout.log; tail -3 $OUT/stdout.log; exit 1; }
  python3 - > gpurun_out/pmcset_${TAG}_$S.txt <<PY
import csv, re, collections
rows=list(csv.DictReader(open("$OUT/p_counter_collection.csv")))
by=collections.OrderedDict()
for r in rows:
    m=re.search(r"\b(k_[a-z_0-9]+)(?:<[^>]*>)?\s*\(", r["Kernel_Name"])
    if not m: continue
    d=by.setdefault(int(r["Dispatch_Id"]), {"k":m.group(1), "dur":(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000})
    d[r["Counter_Name"]]=d.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
ids=sorted(by)
g=[i for i in ids if by[i]["k"]=="k_generate"][-2]
for i in ids:
    if i<g: continue
    d=by[i]
    if d["k"]=="k_generate" and i!=g: break
    print(d["k"], "dur_us=%.0f"%d["dur"], " ".join("%s=%.0f"%(k,v) for k,v in d.items() if k not in ("k","dur")))
PY
  echo "== $S"; cat gpurun_out/pmcset_${TAG}_$S.txt
done
