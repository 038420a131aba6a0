#!/bin/bash
# SQ counters of the bf16-pipe weight-gradient kernel on one shape: tools/debug/pmc_wgrad_split.sh <tag> <ONLY substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
export ONLY="$2" MODES=split3 REPS=2
O=$R/gpurun_out/pmc_ws_$tag
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/debug/wgrad_split_times.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/debug/wgrad_split_times.py > $O/p2.log 2>&1
python3 - <<P
import csv, glob, collections
for p in ("p1", "p2"):
    c = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "wgrad_split" in r["Kernel_Name"]:
                c[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(c): print(p, k, "%.4g" % (c[k] / max(n[k], 1)), "per dispatch over", n[k])
P
