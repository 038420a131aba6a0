cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -n 3
for i in 1 2; do
for w in 0 1; do
  echo "== wide=$w"; CASAPOSE_GEMM_WIDE=$w python tools/debug/gemm_f16x2_probe.py 2>/dev/null | tr '\n' ';'; echo
done; done
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
for w in 0 1; do echo -n "wide=$w "; CASAPOSE_GEMM_WIDE=$w timeout 300 python bench.py $Q 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done
done
