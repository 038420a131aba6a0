#!/usr/bin/env python3
"""cp_wino_gemm_split_planes_f32 with 3 and 2 operand planes on the Winograd GEMM shapes of the bs-16 forward: what half the products buy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
from casapose_amd.engine import split_wino_weights

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for planes in (3, 2):
    tot = 0.0
    for name, tp, k, n, cnt in [("stage4 512->512 (x3)", 5120, 512, 512, 3), ("stage4 256->512", 5120, 256, 512, 1), ("block1 512->256", 4864, 512, 256, 1),
                                ("stage3 256->256 (x3)", 5120, 256, 256, 3), ("stage3 128->256", 5120, 128, 256, 1), ("block2 384->128", 4864, 384, 128, 1),
                                ("stage2 128->128 (x3)", 4864, 128, 128, 3)]:
        rows = 36 * tp
        V = torch.randn(rows, k, device=dev)
        U = torch.randn(36, n, k, device=dev)
        M = torch.empty(rows, n, device=dev)
        Us = split_wino_weights(U, 36, n, k)
        f = lambda: check(lib.cp_wino_gemm_split_planes_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, k, n, planes, st), "split")
        f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tot += ms * cnt
        print("planes %d %-24s %7.3f ms" % (planes, name, ms))
    print("planes %d: sum over the 13 Winograd GEMMs of a step: %.3f ms" % (planes, tot))
