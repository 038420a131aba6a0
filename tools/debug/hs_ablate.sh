# per-layer times of the forward plan with conv_hsplit timing variants (results are garbage, times are not): which side of the kernel bounds a tile
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O; export VARIANTS
for v in ${VARIANTS:-base HS_LOADER_IDLE HS_CONSUMER_IDLE HS_NOINTERP HS_NOEPI HS_NOLOAD}; do
  if [ $v = base ]; then unset CASAPOSE_HIP_LIB; else export CASAPOSE_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
  CASAPOSE_F16X2_GUARD=0 python tools/layer_times.py --reps 10 2>/dev/null | grep -v "^stage3\|^stage4" > $O/lt_$v.txt
done
unset CASAPOSE_HIP_LIB
python - <<'PY'
import glob, os
tabs = {}
for f in sorted(glob.glob("gpurun_out/r05b/lt_*.txt")):
    v = os.path.basename(f)[3:-4]
    for ln in open(f):
        p = ln.split()
        if len(p) >= 6 and p[1].startswith("P"):
            tabs.setdefault(p[0], {})[v] = float(p[-2])
vs = os.environ.get("VARIANTS", "base HS_LOADER_IDLE HS_CONSUMER_IDLE HS_NOINTERP HS_NOEPI HS_NOLOAD").split()
print("%-30s" % "layer" + "".join("%18s" % v for v in vs))
for n, t in tabs.items():
    print("%-30s" % n + "".join("%18.3f" % t.get(v, float("nan")) for v in vs))
PY
