#!/usr/bin/env python3
"""one Winograd plane-set GEMM shape, repeated (for rocprofv3 --pmc): gemm_one.py [K] [N] [reps] [zero]; CASAPOSE_GEMM_ONE=f16x2 selects the fp16 two-way split"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
from casapose_amd.engine import split_wino_weights, split_wino_weights_f16x2
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
k = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tp = 5120; rows = 36 * tp
V = torch.randn(rows, k, device=dev).relu_() if len(sys.argv) <= 4 else torch.zeros(rows, k, device=dev)
U = torch.randn(36, n, k, device=dev); M = torch.empty(rows, n, device=dev)
if os.environ.get("CASAPOSE_GEMM_ONE", "") == "f16x2":
    Us, cs = split_wino_weights_f16x2(U, 36, n, k)
    for _ in range(reps):
        check(lib.cp_wino_gemm_split_scaled_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, k, n, _lib.PLANES_F16X2, cs, st), "f16x2")
else:
    Us = split_wino_weights(U, 36, n, k)
    for _ in range(reps):
        check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, k, n, st), "split")
torch.cuda.synchronize()
