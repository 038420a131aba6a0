set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_hsplit_head.so timeout 300 python bench.py $Q > $O/bench_head_$i.json 2>$O/bench.err
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_issuefirst.so timeout 300 python bench.py $Q > $O/bench_issuefirst_$i.json 2>>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_storesfirst_$i.json 2>>$O/bench.err
done
timeout 300 python tools/layer_times.py > $O/layer_times.txt 2>&1
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_hsplit_head.so timeout 300 python tools/layer_times.py > $O/layer_times_head.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_hsplit.py tests/test_gpu_forward.py tests/test_golden.py -m gpu -x -q > $O/tests.txt 2>&1
tail -n 4 $O/tests.txt
grep -o '"value": [0-9.]*' $O/bench_*.json
