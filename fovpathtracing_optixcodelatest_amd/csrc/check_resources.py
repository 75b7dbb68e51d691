#!/usr/bin/env python3
"""Build guard (Makefile): reads the -Rpass-analysis=kernel-resource-usage remarks of a device compile and fails when a
per-frame kernel (k_generate, k_traverse, k_shade, k_resolve) uses scratch memory, or when k_traverse drops below the waves per
SIMD it is written for.  (The one-off kernels of the BVH build -- k_collapse4, rocPRIM's radix sort -- do use scratch, and run.)

Why: every hot kernel here is tuned to a register budget; a compiler or flag change that makes one spill would silently
turn LDS / register traffic into scratch traffic.  (Round 2 saw a diagnostic build of k_traverse that used scratch die with
HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; the product must never get there unnoticed.)"""
import re, sys

HOT = ("k_generate", "k_traverse", "k_shade", "k_resolve")
text = open(sys.argv[1]).read()
bad = []
name = None
for line in text.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        continue
    m = re.search(r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)", line)
    if m and name and any(h in name for h in HOT) and int(m.group(1)) != 0:
        bad.append("%s uses %s bytes of scratch per lane" % (name, m.group(1)))
    m = re.search(r"remark:\s+Occupancy \[waves/SIMD\]: (\d+)", line)
    if m and name and "k_traverse" in name and int(m.group(1)) < 8 and "--allow-low-occupancy" not in sys.argv:
        bad.append("%s reaches only %s waves per SIMD (8 expected)" % (name, m.group(1)))
if bad:
    sys.stderr.write("check_resources: " + "; ".join(bad) + "\n")
    sys.exit(1)
