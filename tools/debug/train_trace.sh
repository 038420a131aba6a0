#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py --mode train` (environment of the caller, e.g. CASAPOSE_CONV_MODE=bf16) + the per-op table, outside a profile round
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/train_trace -o train --output-format csv -- python3 $R/bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_profiled.json 2> $O/rocprof_train.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python3 $R/tools/train_times.py > $O/train_times.txt 2>&1
ls $O/train_trace
