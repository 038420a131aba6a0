#!/bin/bash
# rocprofv3 kernel-trace stats of the training bench -> gpurun_out/prof_train_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r01}
O=$R/gpurun_out/prof_train_$tag
mkdir -p $O
python3 $R/bench.py --mode train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o train --output-format csv -- python3 $R/bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_profiled.json 2> $O/rocprof.err
find $O/trace -name "*kernel_stats.csv" | head
cat $O/bench_train.json
