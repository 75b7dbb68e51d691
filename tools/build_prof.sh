#!/bin/bash
# kernel times of the hierarchy build alone: tools/build_prof.sh <atrium|street> <triangles> -> gpurun_out/buildprof_<scene>.txt
S=${1:-street}; N=${2:-3800000}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/build_only.py <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from fovpathtracing_optixcodelatest_amd import renderer, scenes
m = scenes.street($N, material="app") if "$S" == "street" else scenes.atrium($N)
r = renderer.SampleRenderer(m)
print("build ms", r.stats().ms_bvh_build, "nodes", r.stats().num_bvh_nodes)
PY
OUT=gpurun_out/buildprof_$S
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 /tmp/build_only.py > $OUT/stdout.log 2>&1 || { echo FAILED; tail -5 $OUT/stdout.log; exit 1; }
grep "build ms" $OUT/stdout.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/b_kernel_stats.csv")))
for r in rows[:14]:
    n = r["Name"]; n = n[n.find("k_"):][:40] if "k_" in n else n[:60]
    print("%-42s calls %4s total_ms %8.2f avg_us %9.1f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
