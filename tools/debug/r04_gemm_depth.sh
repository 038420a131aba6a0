cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -n 2
for i in 1 2; do
for v in depth1 base; do
  if [ $v = base ]; then unset CASAPOSE_HIP_LIB; else export CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
  echo "== $v"; python tools/debug/gemm_f16x2_probe.py 2>/dev/null | tr '\n' ';'; echo; python tools/debug/gemm_split_probe.py 2>/dev/null | tail -n 1
done; done
unset CASAPOSE_HIP_LIB
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_depth1.so timeout 300 python bench.py $Q 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
timeout 300 python bench.py $Q 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
done
