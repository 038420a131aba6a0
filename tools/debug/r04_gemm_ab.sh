set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
for i in 1 2; do
python tools/debug/gemm_split_probe.py > $O/gemm_new$i.txt 2>&1
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_rowsconsec.so python tools/debug/gemm_split_probe.py > $O/gemm_old$i.txt 2>&1
done
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_rowsconsec.so timeout 300 python bench.py $Q > $O/bench_old$i.json 2>$O/bench.err
timeout 300 python bench.py $Q > $O/bench_new$i.json 2>>$O/bench.err
done
tail -n 1 $O/gemm_*.txt; grep -o '"value": [0-9.]*' $O/bench_*.json
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "split or winograd or gemm" 2>&1 | tail -n 2
