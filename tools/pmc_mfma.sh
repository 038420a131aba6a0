#!/bin/bash
# rocprof-reported MFMA utilisation per kernel of the default bench: one rocprofv3 --pmc pass (kernel-trace only) -> gpurun_out/pmc_mfma_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r01}
O=$R/gpurun_out/pmc_mfma_$tag
mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/run.log 2>&1
python3 $R/tools/pmc_mfma_parse.py $O > $O/summary.txt 2>&1
cat $O/summary.txt | head -30
