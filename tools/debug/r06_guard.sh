#!/bin/bash
# round 6: the range guard behind the C ABI -- parity tests + what arming costs (A/B in one call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_guard; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_forward.py tests/test_gpu_hsplit.py tests/test_gpu_dp.py -m gpu -q 2>&1 | tail -40 > $O/tests.txt
cat $O/tests.txt
if [ "$1" = "ab" ]; then bash tools/debug/ab_bench.sh CASAPOSE_F16X2_MONITOR=0 CASAPOSE_F16X2_MONITOR=1 CASAPOSE_F16X2_MONITOR=0 CASAPOSE_F16X2_MONITOR=1 2>&1 | tee $O/ab.txt; fi
