#!/bin/bash
# training: gradient tests in the new default (f16x2 forward, monitored) and the step time beside the exact forward, one call
cd $GRAFT_REPO_ROOT
O=gpurun_out/train_ab; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py tests/test_gpu_scripts.py -m gpu -q -x 2>&1 | tail -8 | tee $O/tests.txt
for v in ${VARIANTS:-"-" "CASAPOSE_TRAIN_FWD=split" "-" "CASAPOSE_TRAIN_FWD=split"}; do
  ( if [ "$v" != "-" ]; then export $v; fi
    echo -n "[$v] "; python bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(p['value'], p['ms_per_step'], p['losses'])" )
done | tee $O/ab.txt
