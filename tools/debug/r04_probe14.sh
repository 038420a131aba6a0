set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_one_object.py -m gpu -x -q > $O/tests_train.txt 2>&1
tail -n 4 $O/tests_train.txt
CASAPOSE_STEM_SPLIT=0 timeout 300 python bench.py --mode train --steps 8 --warmup 3 > $O/train_stemf32.json 2>$O/bench.err
timeout 300 python bench.py --mode train --steps 8 --warmup 3 > $O/train_stemsplit.json 2>>$O/bench.err
CASAPOSE_CONV_MODE=bf16 timeout 300 python bench.py --mode train --steps 8 --warmup 3 > $O/train_bf16.json 2>>$O/bench.err
grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus": 1, "steps": 8, "warmup": 3, "ms_per_step": [0-9.]*' $O/train_*.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
