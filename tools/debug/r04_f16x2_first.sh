set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_hsplit.py -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc $?"
grep -E "max .*rms|network error|passed|failed|Error" $O/tests.log | tail -30
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "every_conv_mode" > $O/tests_fwd.log 2>&1; tail -n 3 $O/tests_fwd.log
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for m in split f16x2 split f16x2; do
CASAPOSE_INFER_CONV_MODE=$m timeout 300 python bench.py $Q > $O/bench_$m.json 2>$O/bench_$m.err; grep -o '"value": [0-9.]*' $O/bench_$m.json | head -1
done
