set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/suite.log 2>&1; echo "suite rc $?"; tail -n 5 $O/suite.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
r=json.load(open("$O/bench.json"))
print(r["value"], r["ms_per_step"], r["dtype"][:80])
print({k:(v["value"] if isinstance(v,dict) and "value" in v else None) for k,v in r.items() if k in ("exact_fp32_mfma","exact_bf16_split","training_leg","training_leg_bf16_convs","cpu_baseline")})
print(r["cpu_baseline"].get("accuracy_vs_fp64"))
rf=r["roofline"]; print({k:rf[k] for k in ("frac","useful_tflops","useful_frac_of_fp32_equiv_peak","fp32_equiv_peak","winograd_transform_ms_per_step","all_conv_ms_per_step")}); print(rf["families"])
PY
