#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of tools/pmc.sh into per-kernel HBM bytes per launch.

Units and corrections follow /opt/skills/guides (MI355X_MICROARCH.md, section HBM, and
cdna_hip_programming.md section 7): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-B read requests at 64 B, so the read side is doubled; WRITE_SIZE is exact for wide
stores.  (The doubling is calibrated for wide coalesced streams; for this path's 16-B-per-lane
gathers it is an upper estimate of the read bytes.)"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_%s_%s" % (tag, counter), "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            m = re.search(r"\b(k_[a-z_0-9]+)(?:<[^>]*>)?\s*\(", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"][:40]
            acc[name][0] += float(row["Counter_Value"])
            acc[name][1] += 1
    res[counter] = {k: (v[0], v[1]) for k, v in acc.items()}
out = {}
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    f, nf = res["FETCH_SIZE"].get(k, (0.0, 0))
    w, nw = res["WRITE_SIZE"].get(k, (0.0, 0))
    if not k.startswith("k_"):
        continue
    fetch = 2.0 * f * 1024.0 / max(1, nf)
    write = w * 1024.0 / max(1, nw)
    out[k] = {"launches": nf, "fetch_bytes_per_launch_x2_corrected": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write}
path = os.path.join(root, "pmc_traffic_%s.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
