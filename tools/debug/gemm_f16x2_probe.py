#!/usr/bin/env python3
"""the f16x2 split GEMM on the Winograd shapes of the bs-16 forward (timing; CASAPOSE_HIP_LIB selects experiment builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
from casapose_amd.engine import split_wino_weights_f16x2
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
tot = 0.0
for name, tp, k, n, cnt in [("512->512", 5120, 512, 512, 3), ("256->512", 5120, 256, 512, 1), ("512->256", 4864, 512, 256, 1), ("256->256", 5120, 256, 256, 3),
                            ("128->256", 5120, 128, 256, 1), ("384->128", 4864, 384, 128, 1)]:
    rows = 36 * tp
    V = torch.randn(rows, k, device=dev).relu_(); U = torch.randn(36, n, k, device=dev); M = torch.empty(rows, n, device=dev)
    Us, cs = split_wino_weights_f16x2(U, 36, n, k)
    f = lambda: check(lib.cp_wino_gemm_split_scaled_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, k, n, _lib.PLANES_F16X2, cs, st), "f16x2")
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10; tot += ms * cnt
    print("%-10s %7.3f ms" % (name, ms))
print("sum over the 10 Winograd GEMMs of an f16x2 step: %.3f ms" % tot)
