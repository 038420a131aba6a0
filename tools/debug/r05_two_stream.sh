# single-stream vs two-stream forward on the current tree over block counts / skews / modes (tools/debug/two_stream_bench.py per setting)
cd $GRAFT_REPO_ROOT
for e in "CASAPOSE_TWO_STREAM_BLOCKS=224" "CASAPOSE_TWO_STREAM_BLOCKS=192" "CASAPOSE_TWO_STREAM_BLOCKS=256" "CASAPOSE_TWO_STREAM_BLOCKS=128" "CASAPOSE_TWO_STREAM_MODE=tag" "CASAPOSE_TWO_STREAM_SKEW=0.2" "CASAPOSE_TWO_STREAM_SKEW=0.6"; do
  ( export $e; echo "[$e]"; timeout 300 python tools/debug/two_stream_bench.py 2>&1 | tail -2 )
done
