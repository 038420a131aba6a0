#!/bin/bash
# the whole -m gpu suite + the default bench line of the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/full_suite; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/full_suite/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "bs32", d.get("roofline_bs32"))
print("f32", d.get("exact_fp32_mfma", {}).get("value"), "split", d.get("exact_bf16_split", {}).get("value"), "train", d.get("training_leg", {}).get("ms_per_step"), d.get("training_leg_bf16_convs", {}).get("ms_per_step"))
print("guard", d["config"].get("f16x2_guard")); print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("accuracy_vs_fp64"))
print("comm_structure", d.get("training_leg", {}).get("comm_structure"))
PY
