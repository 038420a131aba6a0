set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "fused_winograd" > $O/tests_wino.txt 2>&1
tail -n 5 $O/tests_wino.txt
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CP_WINO_OUT_IN_MIN_QUADS=8 timeout 300 python bench.py $Q > $O/bench_q8_$i.json 2>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_q4_$i.json 2>>$O/bench.err
done
timeout 300 python tools/layer_times.py > $O/layer_times.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_golden.py tests/test_gpu_dp.py -m gpu -x -q > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
grep -ho '"value": [0-9.]*' $O/bench_*.json
