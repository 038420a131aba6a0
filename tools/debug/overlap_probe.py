#!/usr/bin/env python3
"""Does an HBM-bound Winograd transform pair overlap the split GEMM when the GEMM leaves CUs free?  Stage-4 shapes, bs 16 halves (8 images)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
from casapose_amd.engine import split_wino_weights
lib = _lib.load(); dev = torch.device("cuda:0")
b, h, w, dil, c = 8, 60, 80, 4, 512
t, tp = C.c_int(0), C.c_int(0); check(lib.cp_wino_tiles(b, h, w, dil, C.byref(t), C.byref(tp))); tp = tp.value
rows = 36 * tp
V = torch.randn(rows, c, device=dev); U = torch.randn(36, c, c, device=dev); M = torch.empty(rows, c, device=dev); Us = split_wino_weights(U, 36, c, c)
x = torch.randn(b, h, w, c, device=dev); y = torch.empty_like(x); V2 = torch.empty(rows, c, device=dev); M2 = torch.randn(rows, c, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(st): check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, c, c, st))
def trans(st):
    check(lib.cp_wino_output_transform_f32(M2.data_ptr(), c, b, h, w, dil, None, c, None, None, None, 0, y.data_ptr(), c, None, c, st))
    check(lib.cp_wino_input_transform_f32(x.data_ptr(), c, c, b, h, w, dil, V2.data_ptr(), c, 0, st))
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
g = timed(lambda: gemm(sa.cuda_stream)); tr = timed(lambda: trans(sb.cuda_stream))
both = timed(lambda: (gemm(sa.cuda_stream), trans(sb.cuda_stream)))
print("persist blocks %s: GEMM alone %.3f ms, transforms alone %.3f ms, both concurrently %.3f ms (sum %.3f)" % (os.environ.get("CASAPOSE_PERSIST_BLOCKS", "256"), g, tr, both, g + tr))
