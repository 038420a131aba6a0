#!/bin/bash
# the training heads' kernels alone (time and bytes per launch: partly written records against whole lines), then the training step with the two
# slice writers (CASAPOSE_HEAD_RECORDS=0) and with whole records (default), in one call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python3 tools/debug/head_probe.py 2>&1 | grep -v amdgpu.ids
for v in 0 1 0 1; do
  echo "== CASAPOSE_HEAD_RECORDS=$v"; CASAPOSE_HEAD_RECORDS=$v python3 bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
} > gpurun_out/head_ab.txt 2>&1
cat gpurun_out/head_ab.txt
