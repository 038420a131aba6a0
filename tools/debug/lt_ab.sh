#!/bin/bash
# per-layer table of the current tree (+ optional env A/B given as arguments, "-" = defaults) and the quick parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/lt_ab; mkdir -p $O
if [ "${TESTS:-1}" = "1" ]; then timeout 1200 python -m pytest tests/test_gpu_hsplit.py tests/test_gpu_forward.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -5; fi
for v in "${@:--}"; do
  ( if [ "$v" != "-" ]; then export $v; fi
    echo "== [$v]"; python tools/layer_times.py --reps 10 2>/dev/null | grep -v "^stage3\|^stage4\|^pv_block_1_\|^pv_block_2_" )
done | tee $O/lt.txt
