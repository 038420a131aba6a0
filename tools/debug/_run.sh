for m in split f32; do CASAPOSE_WINO_WGRAD=$m python bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])"; done
CASAPOSE_CONV_MODE=bf16 python bench.py --mode train --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 mode', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -q -x -k "matches_autograd or bf16 or directional" 2>&1 | tail -3
