#!/usr/bin/env python3
"""Rate of the persistent grouped GEMM (cp_wino_gemm_f32) on the GEMM shapes of the direct convolutions -- the ceiling a persistent,
cross-tile pipelined version of conv_f32_kernel could approach (same MFMA core, no gather)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for name, rows, n, k in [("stage2 3x3", 76800, 128, 1152), ("stage3_unit1_conv1", 76800, 256, 1152), ("pv_block_6", 76800, 256, 4608), ("pv_block_7", 76800, 128, 3456),
                         ("stage4 sc 1x1", 76800, 512, 256), ("train stage2 (bs32 112x112)", 401408, 128, 1152), ("wino stage4 plane set", 36 * 4864, 512, 512)]:
    V = torch.randn(rows, k, device=dev)
    U = torch.randn(n, k, device=dev)
    M = torch.empty(rows, n, device=dev)
    group = rows if rows % 64 == 0 else None
    for _ in range(2):
        check(lib.cp_wino_gemm_f32(V.data_ptr(), U.data_ptr(), M.data_ptr(), rows, group, k, n, st), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        check(lib.cp_wino_gemm_f32(V.data_ptr(), U.data_ptr(), M.data_ptr(), rows, group, k, n, st), "gemm")
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    line = "%-32s M=%7d N=%4d K=%5d  %7.3f ms  %6.1f TF/s" % (name, rows, n, k, ms, 2.0 * rows * n * k / ms / 1e9)
    if rows % 128 == 0:  # the split-bf16 kernel (fp32-equivalent) on the same problem, and its deviation from the fp32-MFMA result
        M2 = torch.empty_like(M)
        from casapose_amd.engine import split_wino_weights
        Us = split_wino_weights(U, 1, n, k)
        check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Us.data_ptr(), M2.data_ptr(), rows, group, k, n, st), "split")
        e0.record()
        for _ in range(5):
            check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Us.data_ptr(), M2.data_ptr(), rows, group, k, n, st), "split")
        e1.record(); e1.synchronize()
        ms2 = e0.elapsed_time(e1) / 5
        ref = (V[:4096].double() @ U.double().T)
        e_f32 = float((M[:4096].double() - ref).abs().max() / ref.abs().max())
        e_spl = float((M2[:4096].double() - ref).abs().max() / ref.abs().max())
        line += "   | split-bf16: %7.3f ms %6.1f TF/s-eq   max err / range: fp32 MFMA %.2e, split %.2e" % (ms2, 2.0 * rows * n * k / ms2 / 1e9, e_f32, e_spl)
    print(line)
