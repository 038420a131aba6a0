#!/usr/bin/env python3
"""Is the f16x2 split GEMM power-limited?  The same launch on random operands, on zero activations and on zero weights (identical instruction
streams and memory traffic; only the switching activity of the multipliers differs), and the exact bf16 split beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
from casapose_amd.engine import split_wino_weights, split_wino_weights_f16x2
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
tp, k, n = 5120, 512, 512
rows = 36 * tp
M = torch.empty(rows, n, device=dev)
def t(f):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20
for name, V, U in (("random V, random U", torch.randn(rows, k, device=dev).relu_(), torch.randn(36, n, k, device=dev)),
                   ("ZERO V, random U", torch.zeros(rows, k, device=dev), torch.randn(36, n, k, device=dev)),
                   ("random V, ZERO U", torch.randn(rows, k, device=dev).relu_(), torch.zeros(36, n, k, device=dev)),
                   ("V = 1.0, U = 1.0 (low parts zero)", torch.ones(rows, k, device=dev), torch.ones(36, n, k, device=dev))):
    Uh, cs = split_wino_weights_f16x2(U + 0, 36, n, k) if float(U.abs().max()) > 0 else (torch.zeros(lib.cp_wino_split_weights_bytes(36, n, k), dtype=torch.uint8, device=dev), 1.0)
    Ub = split_wino_weights(U, 36, n, k)
    a = t(lambda: check(lib.cp_wino_gemm_split_scaled_f32(V.data_ptr(), Uh.data_ptr(), M.data_ptr(), rows, tp, k, n, _lib.PLANES_F16X2, cs, st), "h"))
    b = t(lambda: check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Ub.data_ptr(), M.data_ptr(), rows, tp, k, n, st), "b"))
    print("%-36s f16x2 %.3f ms   exact bf16 split %.3f ms" % (name, a, b))
