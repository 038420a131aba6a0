cd $GRAFT_REPO_ROOT
for e in "X=1" "CASAPOSE_INFER_CONV_MODE=split" "CASAPOSE_INFER_CONV_MODE=f32" "CASAPOSE_TWO_STREAM_BLOCKS=256"; do
  ( export $e; echo "[$e]"; TRIALS=6 REPS=4 timeout 400 python tools/debug/two_stream_equal.py 2>&1 | tail -12 )
done
