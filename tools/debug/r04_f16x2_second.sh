set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_hsplit.py -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc $?"
grep -E "sx .*max|network error|passed|failed|Error" $O/tests.log | tail -30
CASAPOSE_INFER_CONV_MODE=f16x2 timeout 300 python tools/layer_times.py > $O/layers_f16x2.txt 2>&1
CASAPOSE_INFER_CONV_MODE=split timeout 300 python tools/layer_times.py > $O/layers_split.txt 2>&1
paste <(awk '{print $1, $(NF-1)}' $O/layers_split.txt) <(awk '{print $(NF-1)}' $O/layers_f16x2.txt) | head -50
tail -n 3 $O/layers_f16x2.txt
