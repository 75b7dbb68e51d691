#!/bin/bash
# kernel timeline of rank 0's shard of an N-rank frame on one GPU: tools/shard_timeline.sh N
N=${1:-8}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/shard_run.py <<PY
import sys, os
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
W, H = 1920, 1080
r = renderer.SampleRenderer(scenes.atrium(262144)); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5)).BuildCDF())
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
cfg.rank, cfg.world = 0, $N
r.config = cfg
r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
for _ in range(${FRAMES:-8}):
    r.launchParams.frame.subframe_index = 0; r.render_async()
r.synchronize()
PY
OUT=gpurun_out/prof_shard$N
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o s -- python3 /tmp/shard_run.py > $OUT/stdout.log 2>&1
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$OUT/s_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_generate" in r["Kernel_Name"]]
i0=idx[-2-${SPAN:-1}]; i1=idx[-2]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1+1]:
    m=re.search(r"(k_[a-z_]+)",r["Kernel_Name"])
    print("%-12s q%s start %5.0f end %5.0f dur %4.0f"%(m.group(1) if m else r["Kernel_Name"][:12], r["Queue_Id"], (int(r["Start_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000))
PY
