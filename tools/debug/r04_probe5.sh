set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "fused_winograd or winograd" > $O/tests_wino.txt 2>&1
tail -n 5 $O/tests_wino.txt
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_WINO_FUSE_OUT_IN=0 timeout 300 python bench.py $Q > $O/bench_sep$i.json 2>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_fused$i.json 2>>$O/bench.err
done
timeout 300 python tools/layer_times.py > $O/layer_times.txt 2>&1
CASAPOSE_WINO_FUSE_OUT_IN=0 timeout 300 python tools/layer_times.py > $O/layer_times_sep.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_golden.py -m gpu -x -q > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
grep -ho '"value": [0-9.]*' $O/bench_*.json
