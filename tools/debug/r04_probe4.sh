set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hsplit.py -m gpu -x -q > $O/tests_hsplit.txt 2>&1
tail -n 3 $O/tests_hsplit.txt
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_MATERIALISE_BILINEAR=1 timeout 300 python bench.py $Q > $O/bench_mat$i.json 2>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_fused$i.json 2>>$O/bench.err
done
timeout 300 python tools/layer_times.py > $O/layer_times.txt 2>&1
CASAPOSE_MATERIALISE_BILINEAR=1 timeout 300 python tools/layer_times.py > $O/layer_times_mat.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.txt 2>&1
tail -n 3 $O/tests.txt
grep -ho '"value": [0-9.]*' $O/bench_*.json
