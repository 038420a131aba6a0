#!/usr/bin/env python3
"""Fill the @@NAME@@ placeholders of DESIGN.md from the stamped profile set profiles/<tag>_* (one-off helper of the round's close-out; DESIGN.md
carries ONE set of numbers and they all come from that set).  usage: python tools/fill_design.py r05 [--check]"""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda n: os.path.join(R, "profiles", "%s_%s" % (tag, n))
b = json.load(open(P("bench.json")))
v = json.load(open(P("bench_vote.json")))
rf = b["roofline"]
lt = {}
for ln in open(P("layer_times.txt")):
    p = ln.split()
    if len(p) >= 6 and p[0].startswith(("pv_", "stage", "conv0")):
        i = [k for k, t in enumerate(p) if "x" in t and t[0].isdigit()][0]
        lt[p[0]] = float(p[i + 1] if not p[i].endswith("x") else p[i + 2])   # the ms column follows "N x K"
traffic = json.load(open(P("pmc_traffic.json")))
steps = 3.0   # the PMC passes run `--steps 3 --warmup 1` + one calibration-free forward per roofline-less run: launches / 5 forwards
def fam_gb(prefixes):
    tot = 0.0
    for name, t in traffic.items():
        if any(q in name for q in prefixes):
            tot += t["hbm_bytes_per_launch"] * t["launches"]
    return tot
# per step: each family's average bytes per launch x its launches per bs-16 forward (10 Winograd layers: nine GEMMs in the wide kernel, one in the narrow
# one; seven plain output transforms, eight input transforms, three fused output -> input transforms) -- the sum the round-4 verdict formed
wino_gb = 0.0
for q, per_fwd in (("wino_gemm_wide", 9), ("wino_gemm_split", 1), ("wino_out_kernel", 7), ("wino_in_kernel", 8), ("wino_out_in_kernel", 3)):
    rows = [t for n, t in traffic.items() if q in n]
    if rows:
        n_l = sum(t["launches"] for t in rows)
        wino_gb += sum(t["hbm_bytes_per_launch"] * t["launches"] for t in rows) / n_l * per_fwd / 1e9
acc = b["cpu_baseline"]["accuracy_vs_fp64"]
fams = rf["families"]
hs = fams["conv_hsplit_kernel<2>"]
busy = [ln for ln in open(P("pmc_mfma.txt")) if "conv_hsplit_kernel" in ln]
busy_vals = sorted(float(ln.split()[-2]) for ln in busy)
c = b["cpu_baseline"]
rs = v["ransac"]
sub = {
    "HEAD": "%.0f" % b["value"], "HEADMS": "%.2f" % b["ms_per_step"], "F32": "%.0f" % b["exact_fp32_mfma"]["value"], "SPLIT": "%.0f" % b["exact_bf16_split"]["value"],
    "HSMS": "%.2f" % hs["ms"], "HSTF": "%.0f" % (957.0 / hs["ms"]),
    "B5": "%.3f" % lt["pv_block_5_conv2d"], "B10": "%.3f" % lt["pv_block_10_prepare_conv2d"], "B4": "%.3f" % lt["pv_block_4_conv2d"], "B9": "%.3f" % lt["pv_block_9_prepare_conv2d"],
    "ACC": "f16x2 %.1e / %.1e, fp32 MFMA %.1e / %.1e, exact split %.1e / %.1e (logits / vector field)" % (
        acc["per_conv_mode"]["f16x2"], acc["vector_field_per_conv_mode"]["f16x2"], acc["per_conv_mode"]["f32"], acc["vector_field_per_conv_mode"]["f32"],
        acc["per_conv_mode"]["split"], acc["vector_field_per_conv_mode"]["split"]),
    "TRAIN": "%.1f" % b["training_leg"]["ms_per_step"], "TRAINBF": "%.1f" % b["training_leg_bf16_convs"]["ms_per_step"],
    "CPU": "%.1f" % c["value"], "RANSAC": "%.2f" % rs["ms_per_call"], "RTESTS": "%.2e" % rs["cosine_tests_per_s"], "RVALU": "%.2f" % rs["valu"]["frac"],
    "WINOMS": "%.2f" % rf["winograd"]["ms_per_step"], "WGEMM": "%.2f" % rf["winograd"]["gemm_ms"], "WTR": "%.2f" % rf["winograd"]["transform_ms"],
    "STEM": "%.2f" % fams["conv_stem_split_kernel"]["ms"], "VOTE": "%.2f" % v["ms_per_step"], "WTRAF": "%.1f" % wino_gb,
    "HSBUSY": "%.0f-%.0f %% MFMA-busy at 2.2-2.4 GHz over its instantiations" % (busy_vals[0], busy_vals[-1]),
    "FRAC": "%.3f" % rf["frac"], "USEFUL": "%.0f" % rf["useful_tflops"], "USEFULF": "%.2f" % rf["useful_frac_of_fp32_equiv_peak"],
    "STEPSPLIT": "%.2f ms per step = %.2f `conv_hsplit` + %.2f Winograd GEMMs + %.2f Winograd transforms + %.2f stem + %.2f fp32-MFMA layers + ~%.2f voter, pooling, label pyramid and launch gaps" % (
        b["ms_per_step"], hs["ms"], rf["winograd"]["gemm_ms"], rf["winograd"]["transform_ms"], fams["conv_stem_split_kernel"]["ms"], fams["conv_f32_kernel<2,2,1,1><64x64>"]["ms"],
        b["ms_per_step"] - rf["all_conv_ms_per_step"]),
}
d = open(os.path.join(R, "DESIGN.md")).read()
missing = sorted(set(re.findall(r"@@([A-Z0-9]+)@@", d)) - set(sub))
assert not missing, missing
for k, val in sub.items():
    d = d.replace("@@%s@@" % k, val)
if "--check" in sys.argv:
    print(json.dumps(sub, indent=1))
else:
    open(os.path.join(R, "DESIGN.md"), "w").write(d)
    print("filled %d placeholders from profiles/%s_*" % (len(sub), tag))
