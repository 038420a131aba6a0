set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
python tools/debug/gemm_split_probe.py > $O/gemm_transposed.txt 2>&1
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_rowmajor.so python tools/debug/gemm_split_probe.py > $O/gemm_rowmajor.txt 2>&1
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_rowmajor.so timeout 300 python bench.py $Q > $O/bench_rowmajor$i.json 2>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_transposed$i.json 2>>$O/bench.err
done
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "split or winograd or gemm" > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
tail -n 2 $O/gemm_*.txt
grep -ho '"value": [0-9.]*' $O/bench_*.json
