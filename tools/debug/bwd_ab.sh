#!/bin/bash
# training backward in the fp16 two-way split: kernel test, the training suites, then the step both ways in one call
cd $GRAFT_REPO_ROOT
O=gpurun_out/bwd_ab; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_wgrad_split.py -m gpu -q -x 2>&1 | tail -4 | tee $O/kernel_tests.txt
VARIANTS="- CASAPOSE_TRAIN_BWD=split - CASAPOSE_TRAIN_BWD=split" bash tools/debug/train_ab.sh
