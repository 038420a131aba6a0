cd $GRAFT_REPO_ROOT
for e in "CASAPOSE_TWO_STREAM_SKEW=0.6" "CASAPOSE_TWO_STREAM_SKEW=0.6" "CASAPOSE_TWO_STREAM_SKEW=0.5" "CASAPOSE_TWO_STREAM_SKEW=0.7" "CASAPOSE_TWO_STREAM_SKEW=0.9" "CASAPOSE_TWO_STREAM_SKEW=0.4"; do
  ( export $e; echo "[$e]"; timeout 300 python tools/debug/two_stream_bench.py 2>&1 | tail -2 )
done
