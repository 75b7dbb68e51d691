#!/bin/bash
# kernel timeline of one steady-state frame (rocprofv3 kernel trace of the bench command)
TAG=${1:-tl}
tools/prof.sh $TAG > gpurun_out/prof.log 2>&1
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_generate" in r["Kernel_Name"]]
i0=idx[-3]; i1=idx[-2]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1+1]:
    m=re.search(r"(k_[a-z_]+)",r["Kernel_Name"])
    print("%-12s q%s grid %7s start %5.0f end %5.0f dur %4.0f"%(m.group(1) if m else r["Kernel_Name"][:12], r["Queue_Id"], r["Grid_Size_X"], (int(r["Start_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-t0)/1000, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000))
PY
