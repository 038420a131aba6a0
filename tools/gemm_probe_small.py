import os, sys
sys.path.insert(0, "/root/repo")
import torch
from casapose_amd import _lib
from casapose_amd._lib import check
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
for name, rows, n, k in [("conv0", 1228800, 64, 224), ("stage1 sc", 307200, 64, 64), ("stage2 sc", 76800, 128, 64), ("stage3 sc", 76800, 256, 128), ("stage4 sc", 76800, 512, 256)]:
    V = torch.randn(rows, k, device=dev); U = torch.randn(n, k, device=dev); M = torch.empty(rows, n, device=dev)
    for _ in range(2): check(lib.cp_wino_gemm_f32(V.data_ptr(), U.data_ptr(), M.data_ptr(), rows, rows, k, n, st), "g")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): check(lib.cp_wino_gemm_f32(V.data_ptr(), U.data_ptr(), M.data_ptr(), rows, rows, k, n, st), "g")
    e1.record(); e1.synchronize(); ms = e0.elapsed_time(e1) / 10
    print("%-10s M=%8d N=%4d K=%4d  %.3f ms  %.1f TF/s" % (name, rows, n, k, ms, 2.0 * rows * n * k / ms / 1e9))
