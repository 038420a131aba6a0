# A/B inside one call: bench forward-only numbers for env switches given as arguments ("NAME=VAL" pairs separated by spaces, one variant per argument; "-" = defaults)
set -u
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  ( if [ "$v" != "-" ]; then export $v; fi
    echo -n "[$v] "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-leg --no-optin 2>/dev/null | python -c "import json,sys; p=json.loads(sys.stdin.read()); r=p['roofline']; print(p['value'], p['ms_per_step'], 'hsplit', r['families']['conv_hsplit_kernel<2>']['ms'], 'wino gemm', r['winograd']['gemm_ms'], 'transforms', r['winograd']['transform_ms'])" )
done
