set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_HS_TABLES_GLOBAL=1 python bench.py $Q > $O/bench_global$i.json 2>$O/bench.err
python bench.py $Q > $O/bench_lds$i.json 2>>$O/bench.err
done
python tools/layer_times.py > $O/layer_times.txt 2>&1
python -m pytest tests/test_gpu_hsplit.py tests/test_gpu_halo.py tests/test_gpu_forward.py tests/test_golden.py -m gpu -x -q > $O/tests.txt 2>&1
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "bf16_gradient" > $O/tests_bf16grad.txt 2>&1
tail -n 3 $O/tests.txt; tail -n 5 $O/tests_bf16grad.txt
grep -ho '"value": [0-9.]*' $O/bench_*.json
