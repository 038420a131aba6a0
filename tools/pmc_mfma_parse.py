#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from the pass of tools/pmc_mfma.sh:
   util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); effective clock = (GRBM_GUI_ACTIVE / 8) / kernel duration."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
cnt = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            n[k] += 1
# durations joined PER DISPATCH (the round-1 table summed every dispatch of the trace by kernel name while the counter file covered fewer
# of them, which produced impossible clocks for two instantiations): only dispatches that have counter rows contribute their duration
have = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        have[r["Kernel_Name"].replace("(anonymous namespace)::", "")].add(r.get("Dispatch_Id"))
dur = defaultdict(float)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if r.get("Dispatch_Id") in have[k]:
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for k, c in cnt.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0 or c.get("SQ_INSTS_MFMA", 0.0) <= 0:
        continue
    rows.append((dur[k], k, n[k], 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), cyc / max(dur[k], 1e-9) / 1e3))
print("%-78s %6s %10s %9s %9s" % ("kernel (MFMA kernels only)", "calls", "total ms", "MFMA %", "GHz"))
for d, k, calls, util, ghz in sorted(rows, reverse=True):
    flag = "" if 1.0 <= ghz <= 2.45 else "   <- clock outside the part's range: counter / trace rows do not match, not evidence"
    print("%-78s %6d %10.3f %9.1f %9.2f%s" % (k[:78], calls, d / 1e3, util, ghz, flag))
