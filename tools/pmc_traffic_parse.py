#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from the two PMC passes of tools/pmc_traffic.sh.
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half
of the bytes of a wide (16 B/lane) coalesced read (MI355X_MICROARCH.md, HBM section) -- all operand loads here are
16 B/lane buffer loads.  WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            acc[name][0] += float(r["Counter_Value"])
            acc[name][1] += 1
    return acc


def main():
    root = sys.argv[1]
    fetch = per_kernel(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "write"), "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write), key=lambda n: -(fetch.get(n, [0, 1])[0])):
        f, nf = fetch.get(name, [0.0, 0])
        w, nw = write.get(name, [0.0, 0])
        fb = 2.0 * 1024.0 * f / max(nf, 1)
        wb = 1024.0 * w / max(nw, 1)
        out[name] = {"launches": max(nf, nw), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
        print("%-100s n=%5d  read %10.2f MB  write %10.2f MB per launch" % (name[:100], max(nf, nw), fb / 1e6, wb / 1e6))
    json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
