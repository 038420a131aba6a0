#!/usr/bin/env python3
"""cp_wino_gemm_split_f32 on the Winograd plane sets of the bs-16 forward (36 groups x Tp rows), TFLOP/s executed (x6) and fp32-equivalent.
CASAPOSE_GEMM_PREFETCH=1|2|3 selects the producers' prefetch depth (chunks of A in flight)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
from casapose_amd.engine import split_wino_weights

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
tot = 0.0
for name, tp, k, n, cnt in [("stage4 512->512 (x3)", 5120, 512, 512, 3), ("stage4 256->512", 5120, 256, 512, 1), ("block1 512->256", 4864, 512, 256, 1),
                            ("stage3 256->256 (x3)", 5120, 256, 256, 3), ("stage3 128->256", 5120, 128, 256, 1), ("block2 384->128", 4864, 384, 128, 1),
                            ("stage2 128->128 (x3)", 4864, 128, 128, 3)]:
    rows = 36 * tp
    V = torch.randn(rows, k, device=dev)
    U = torch.randn(36, n, k, device=dev)
    M = torch.empty(rows, n, device=dev)
    Us = split_wino_weights(U, 36, n, k)
    for _ in range(2):
        check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, k, n, st), "split")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        check(lib.cp_wino_gemm_split_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), rows, tp, k, n, st), "split")
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms * cnt
    ref = torch.einsum("tk,nk->tn", V[:512].double(), U[0].double())
    err = float((M[:512].double() - ref).abs().max() / ref.abs().max())
    print("%-24s rows=%7d K=%4d N=%4d  %7.3f ms  %7.1f TF/s executed  %6.1f fp32-equivalent   err %.1e" % (name, rows, k, n, ms, 12.0 * rows * n * k / ms / 1e9, 2.0 * rows * n * k / ms / 1e9, err))
print("sum over the 13 Winograd GEMMs of a step: %.3f ms (prefetch depth %s)" % (tot, os.environ.get("CASAPOSE_GEMM_PREFETCH", "default")))
