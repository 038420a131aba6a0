set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x; mkdir -p $O
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for i in 1 2; do
for m in 0 1; do
CASAPOSE_WINO_DIRECT_128=$m timeout 300 python bench.py $Q > $O/bench_$m.json 2>$O/bench_$m.err; echo -n "direct128=$m "; grep -o '"value": [0-9.]*' $O/bench_$m.json | head -1
done; done
CASAPOSE_WINO_DIRECT_128=1 timeout 300 python tools/layer_times.py 2>&1 | grep "stage2\|whole"
