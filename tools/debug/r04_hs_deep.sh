set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hsplit.py tests/test_gpu_f16x2.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -n 2 $O/tests.log
for tile in 102 100; do
echo -n "b4 t$tile: "; python tools/one_layer.py --batch 16 --h 240 --w 320 --cin 128 --cout 32 --dil 1 --tile $tile --reps 20 2>/dev/null | tail -n 1
echo -n "b5 t$tile: "; python tools/one_layer.py --batch 16 --h 480 --w 640 --cin 32 --cout 32 --dil 1 --tile $tile --reps 20 2>/dev/null | tail -n 1
echo -n "b6 t$tile: "; python tools/one_layer.py --batch 16 --h 60 --w 80 --cin 512 --cout 64 --dil 1 --tile $tile --reps 20 2>/dev/null | tail -n 1
echo -n "b3 t$tile: "; python tools/one_layer.py --batch 16 --h 120 --w 160 --cin 192 --cout 64 --dil 1 --tile $tile --reps 20 2>/dev/null | tail -n 1
done
Q="--no-cpu-baseline --no-roofline --no-optin --no-train-leg --steps 30 --warmup 5"
for m in f16x2 split f16x2 split; do
CASAPOSE_INFER_CONV_MODE=$m timeout 300 python bench.py $Q > $O/bench_$m.json 2>$O/bench_$m.err; echo -n "$m "; grep -o '"value": [0-9.]*' $O/bench_$m.json | head -1
done
