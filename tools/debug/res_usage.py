#!/usr/bin/env python3
"""VGPRs / spills / scratch of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage); optional substring filter.
    python tools/debug/res_usage.py conv_hsplit [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "casapose_amd", "csrc", sys.argv[1] + ".hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-munsafe-fp-atomics", "-fno-slp-vectorize",
                    "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", "/dev/null"] + sys.argv[3:], capture_output=True, text=True)
name, rec = None, {}
for ln in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        name = m.group(1); rec[name] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", ln)
    if m and name:
        rec[name][m.group(1).strip()] = int(m.group(2))
for n, d in rec.items():
    if flt in n:
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
        print("%-60s VGPRs %3d  spilled V %3d S %3d  scratch %4d  SGPRs %3d" % (short[:60], d.get("VGPRs", -1), d.get("VGPRs Spill", 0), d.get("SGPRs Spill", 0), d.get("ScratchSize", 0), d.get("TotalSGPRs", 0)))
