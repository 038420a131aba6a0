cd $GRAFT_REPO_ROOT
for v in base WS_NOA WS_NOB WS_NOAB; do
  if [ $v = base ]; then unset CASAPOSE_HIP_LIB; else export CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
  echo "== $v"; python tools/debug/gemm_f16x2_probe.py 2>/dev/null | tr '\n' ';'; echo
done
