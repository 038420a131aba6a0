set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
for d in 1 2 3; do CASAPOSE_GEMM_PREFETCH=$d python tools/debug/gemm_split_probe.py > $O/gemm_d$d.txt 2>&1; done
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 20 --warmup 5"
for d in 1 2 3; do CASAPOSE_GEMM_PREFETCH=$d python bench.py $Q > $O/bench_d$d.json 2>$O/bench_d$d.err; done
python -m pytest tests/test_gpu_voting.py tests/test_gpu_forward.py -m gpu -x -q -k "ccl or ls_voting or filtered" > $O/tests.txt 2>&1
tail -3 $O/gemm_d*.txt; tail -3 $O/tests.txt
grep -ho '"value": [0-9.]*' $O/bench_d*.json
