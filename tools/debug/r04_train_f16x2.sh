set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py tests/test_gpu_scripts.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -n 3 $O/tests.log
for m in f16x2 split f16x2 split; do
CASAPOSE_TRAIN_FWD=$m timeout 600 python bench.py --mode train --steps 8 --warmup 3 > $O/train_$m.json 2>$O/train_$m.err; echo -n "$m "; grep -o '"value": [0-9.]*' $O/train_$m.json | head -1
done
