#!/bin/bash
# tools/r4_bench.sh <tag>: the driver's bench command (defaults, then the driver's own --steps 20 --warmup 5), results under gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
timeout -k 10 700 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench FAILED"; tail -c 1500 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-counters > gpurun_out/${TAG}_bench20.json 2>> gpurun_out/${TAG}_bench.err || { echo "bench20 FAILED"; tail -c 1500 gpurun_out/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
for f in ("gpurun_out/${TAG}_bench.json", "gpurun_out/${TAG}_bench20.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value", d["value"], "ms", d["ms_per_step"], "latency", d.get("frame_latency_ms"), "sync", d.get("value_sync_per_frame"), r["bound"], "l1_frac", r.get("l1_frac"),
          "lane_use", r.get("lane_use"), "hbm", r.get("hbm_frac_measured"), "frac", r["frac"], "frac1", r.get("frac_one_frame_in_flight"), "ser", r.get("frac_serialised"),
          "parity", d.get("parity_vs_oracle_rgba8_mismatch"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    print("  ", {k: v["ms_per_frame"] for k, v in d.get("variants", {}).items()}, d.get("two_chains_per_frame"))
PY
