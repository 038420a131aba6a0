#!/usr/bin/env python3
"""Per-kernel statistics of a device-only assembly listing (hipcc --cuda-device-only -S): lines, exec-mask regions, branches, MFMAs, waits.
usage: isa_stats.py file.s [substring of the mangled kernel name]"""
import re, sys
src, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
name, body, out = None, [], []
for ln in open(src):
    m = re.match(r"^(_Z[\w$]+):", ln)
    if m:
        name, body = m.group(1), []
        continue
    if name is None:
        continue
    body.append(ln)
    if "s_endpgm" in ln:
        if want in name:
            t = "".join(body)
            out.append((name, len(body), t.count("saveexec"), t.count("s_cbranch"), t.count("v_mfma"), t.count("s_waitcnt vmcnt"), t.count("v_readlane") + t.count("v_writelane"), t.count("scratch_")))
        name = None
print("%-86s %6s %8s %7s %5s %6s %6s %7s" % ("kernel", "lines", "saveexec", "branch", "mfma", "vmcnt", "lanes", "scratch"))
for o in out:
    print("%-86s %6d %8d %7d %5d %6d %6d %7d" % o)
