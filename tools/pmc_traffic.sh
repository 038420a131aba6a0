#!/bin/bash
# HBM traffic of the bench kernels: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over
# the default bench command, kernel-trace only (no other trace domains).  -> gpurun_out/pmc_traffic_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r01}
O=$R/gpurun_out/pmc_traffic_$tag
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/write.log 2>&1
ls $O/fetch $O/write
python3 $R/tools/pmc_traffic_parse.py $O > $O/summary.txt 2>&1
cat $O/summary.txt | head -40
