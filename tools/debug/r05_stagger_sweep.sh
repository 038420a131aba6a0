# per-layer times of the forward plan with the conv_hsplit start stagger (CASAPOSE_HS_STAGGER units of s_sleep(32) per phase group)
# (CASAPOSE_HS_STAGGER existed only in the experiment: HSplitK.stagger read from that variable in cp_conv2d_fwd_split, and at the top of the kernel
#  `for (i < ((blockIdx.x >> 3) & 3) * p.stagger) __builtin_amdgcn_s_sleep(32);` -- result in profiles/probes/r05_hsplit_ablation.txt section 3)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; rm -rf $O; mkdir -p $O
VALS=${VALS:-"0 1 2 4 8 0"}
i=0
for b in $VALS; do
  i=$((i+1))
  CASAPOSE_HS_STAGGER=$b CASAPOSE_F16X2_GUARD=0 python tools/layer_times.py --reps 10 2>/dev/null | grep -v "^stage3\|^stage4" > $O/lt_${i}_$b.txt
done
python - <<'PY'
import glob, os
tabs, vs = {}, []
for f in sorted(glob.glob("gpurun_out/r05s/lt_*.txt")):
    v = os.path.basename(f)[3:-4]
    vs.append(v)
    for ln in open(f):
        p = ln.split()
        if len(p) >= 6 and p[1].startswith("P"):
            tabs.setdefault(p[0], {})[v] = float(p[-2])
print("%-30s" % "layer" + "".join("%10s" % v for v in vs))
tot = {v: 0.0 for v in vs}
for n, t in tabs.items():
    print("%-30s" % n + "".join("%10.3f" % t.get(v, float("nan")) for v in vs))
    for v in vs: tot[v] += t.get(v, 0.0)
print("%-30s" % "sum" + "".join("%10.3f" % tot[v] for v in vs))
PY
