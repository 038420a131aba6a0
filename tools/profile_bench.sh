#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r01}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/rocprof.err
ls $O/trace | head
cat $O/bench.json
