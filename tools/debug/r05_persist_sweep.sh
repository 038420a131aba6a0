# per-layer times of the forward plan with fewer persistent blocks per launch: does a tile get faster when fewer CUs run (contention for HBM / L2 in the
# epilogue bursts) or stay the same (issue-bound inside the CU)?
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; rm -rf $O; mkdir -p $O
for b in 256 192 128 64; do
  CASAPOSE_PERSIST_BLOCKS=$b CASAPOSE_F16X2_GUARD=0 python tools/layer_times.py --reps 10 2>/dev/null | grep -v "^stage3\|^stage4" > $O/lt_$b.txt
done
python - <<'PY'
import glob, os
tabs = {}
for f in sorted(glob.glob("gpurun_out/r05p/lt_*.txt")):
    v = os.path.basename(f)[3:-4]
    for ln in open(f):
        p = ln.split()
        if len(p) >= 6 and p[1].startswith("P"):
            tabs.setdefault(p[0], {})[v] = float(p[-2])
vs = ["256", "192", "128", "64"]
print("%-30s" % "layer" + "".join("%10s" % v for v in vs) + "   (ms; then ms x blocks / 256)")
for n, t in tabs.items():
    print("%-30s" % n + "".join("%10.3f" % t.get(v, float("nan")) for v in vs) + "   " + "".join("%8.3f" % (t.get(v, float("nan")) * int(v) / 256) for v in vs))
PY
