#!/bin/bash
# rocprofv3 kernel-trace stats of the voting-stage bench -> gpurun_out/prof_vote_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r01}
O=$R/gpurun_out/prof_vote_$tag
mkdir -p $O
python3 $R/bench.py --mode vote --steps 10 --warmup 3 > $O/bench_vote.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o vote --output-format csv -- python3 $R/bench.py --mode vote --steps 10 --warmup 3 > $O/bench_vote_profiled.json 2> $O/rocprof.err
head -14 $O/trace/vote_kernel_stats.csv | cut -c1-200
