set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hsplit.py tests/test_gpu_forward.py -m gpu -x -q -k "stem or two_stream or forward_in_every or default_inference" > $O/tests.txt 2>&1
tail -n 6 $O/tests.txt
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_STEM_SPLIT=0 timeout 300 python bench.py $Q > $O/bench_stemf32_$i.json 2>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_stemsplit_$i.json 2>>$O/bench.err
done
timeout 300 python tools/layer_times.py 2>&1 | head -6 > $O/layer_times_head.txt; cat $O/layer_times_head.txt
grep -ho '"value": [0-9.]*' $O/bench_*.json
