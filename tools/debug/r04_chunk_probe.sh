set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 20 --warmup 5"
python bench.py $Q > gpurun_out/r04a/base.json 2>gpurun_out/r04a/base.err
for c in 1 2 4 8; do CASAPOSE_WINO_CHUNK=$c python bench.py $Q > gpurun_out/r04a/chunk$c.json 2>gpurun_out/r04a/chunk$c.err; done
python tools/layer_times.py > gpurun_out/r04a/layer_times.txt 2>&1
CASAPOSE_WINO_CHUNK=4 python tools/layer_times.py > gpurun_out/r04a/layer_times_chunk4.txt 2>&1
grep -h images gpurun_out/r04a/*.json | cut -c1-200
