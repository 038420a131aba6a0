"""Winograd weight-gradient GEMM per training shape (bs 32, 448x448): fp32 grouped GEMM (cp_conv2d_wgrad_f32) vs the bf16-pipe kernel
(cp_wino_wgrad_split_f32, exact split / bf16 operands); TF/s are fp32-equivalent (2 * 36 * T * N * K)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from casapose_amd import _lib
from casapose_amd._lib import ConvDesc, check
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
def timed(fn, reps=5):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
T = 6272
for name, n, k in (("stage4 512x512", 512, 512), ("stage4_u1_conv1 512x256", 512, 256), ("stage3 256x256", 256, 256), ("block1 256x512", 256, 512), ("block2 128x384", 128, 384)):
    dm = torch.randn(36, T, n, device=dev); v = torch.randn(36, T, k, device=dev); du = torch.empty(36, n, k, device=dev)
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w = 1, 1, 36 * T, 1, 36 * T
    d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = n, 1, 1, 1, 1, 0
    d.num_sources = 1
    d.src[0].data, d.src[0].channels, d.src[0].ld, d.src[0].mode = v.data_ptr(), k, k, 0
    d.group_rows = T
    fl = 2.0 * 36 * T * n * k
    t32 = timed(lambda: check(lib.cp_conv2d_wgrad_f32(C.byref(d), dm.data_ptr(), n, du.data_ptr(), 0, st)))
    t3 = timed(lambda: check(lib.cp_wino_wgrad_split_f32(dm.data_ptr(), v.data_ptr(), du.data_ptr(), 36, T, n, k, 3, st)))
    t1 = timed(lambda: check(lib.cp_wino_wgrad_split_f32(dm.data_ptr(), v.data_ptr(), du.data_ptr(), 36, T, n, k, 1, st)))
    print("%-26s fp32 %.3f ms %6.1f TF/s | split %.3f ms %6.1f | bf16 %.3f ms %6.1f" % (name, t32, fl / t32 / 1e9, t3, fl / t3 / 1e9, t1, fl / t1 / 1e9))
