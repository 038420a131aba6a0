set -x
python tools/debug/grad_per_variable.py --top 4 > gpurun_out/g_default.log 2>&1
CASAPOSE_CONV_MODE=f32 python tools/debug/grad_per_variable.py --top 4 > gpurun_out/g_conv_f32.log 2>&1
CASAPOSE_CONV_MODE=f32 CASAPOSE_WINO_GEMM=f32 python tools/debug/grad_per_variable.py --top 4 > gpurun_out/g_all_f32.log 2>&1
CASAPOSE_CONV_MODE=f32 CASAPOSE_WINO_GEMM=f32 CASAPOSE_NO_WINOGRAD=1 python tools/debug/grad_per_variable.py --top 4 > gpurun_out/g_all_f32_nowino.log 2>&1
python tools/debug/grad_per_variable.py --top 4 --h 64 --w 64 > gpurun_out/g_default_64.log 2>&1
python tools/debug/grad_per_variable.py --top 4 --h 64 --w 96 --k 9 --b 2 > gpurun_out/g_default_64x96.log 2>&1
python tools/debug/grad_per_variable.py --top 4 --h 96 --w 128 --k 9 --b 2 > gpurun_out/g_default_96x128.log 2>&1
tail -n 14 gpurun_out/g_*.log
python -m pytest tests/test_gpu_train.py -x -q -k "bn_act" 2>&1 | tail -5
