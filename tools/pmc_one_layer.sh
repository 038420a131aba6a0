#!/bin/bash
# usage: tools/pmc_one_layer.sh <tag> [one_layer args...]   (optionally CASAPOSE_HIP_LIB set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
python3 $R/tools/one_layer.py "$@" > $O/plain.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/one_layer.py "$@" --reps 3 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/one_layer.py "$@" --reps 3 > $O/p2.log 2>&1
cat $O/plain.log
