#!/bin/bash
# the training heads' kernels alone, shipped library against variants/lib_BASE.so (the tree before the change), then the training step both ways
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== variants/lib_BASE.so"; python3 tools/debug/head_probe.py variants/lib_BASE.so 2>&1 | grep -v amdgpu.ids
echo "== shipped library"; python3 tools/debug/head_probe.py 2>&1 | grep -v amdgpu.ids
python3 -m pytest tests/test_gpu_train.py -q -x -k "fused_head" 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "== CASAPOSE_HEAD_RECORDS=$v"; CASAPOSE_HEAD_RECORDS=$v python3 bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
} > gpurun_out/head_ab.txt 2>&1
cat gpurun_out/head_ab.txt
python3 -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -q -x 2>&1 | tail -3 | tee -a gpurun_out/head_ab.txt
