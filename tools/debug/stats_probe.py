"""Timing of the batch-statistics routes (cp_bn_stats_f32 alone; cp_wino_output_transform_f32 with and without the fused table) at the training
shapes; CP_BN_STATS_BLOCKS / CP_WINO_STATS_BLOCKS override the grid caps (read once per process)."""
import ctypes as C, os, sys
import torch
sys.path[:0] = ["."]
from casapose_amd import _lib
from casapose_amd._lib import check
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream

def timeit(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("caps: bn", os.environ.get("CP_BN_STATS_BLOCKS"), "wino", os.environ.get("CP_WINO_STATS_BLOCKS"))
for px, c in ((100352, 512), (100352, 256), (401408, 64), (6422528, 32)):
    x = torch.randn(px, c, device=dev)
    sums = torch.zeros(2 * c, dtype=torch.float64, device=dev)
    t = timeit(lambda: check(lib.cp_bn_stats_f32(x.data_ptr(), px, c, c, sums.data_ptr(), st)))
    print("bn_stats  %8d x %3d  %7.1f us  %.2f TB/s" % (px, c, t, px * c * 4 / t / 1e6))
for cout in (512, 256):
    b, h, w = 32, 56, 56
    T, Tp = C.c_int(), C.c_int()
    check(lib.cp_wino_tiles(b, h, w, 4 if cout == 512 else 2, C.byref(T), C.byref(Tp)))
    d = 4 if cout == 512 else 2
    M = torch.randn(36 * Tp.value * cout, device=dev)
    raw = torch.empty(b * h * w * cout, device=dev)
    sums = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    t0 = timeit(lambda: check(lib.cp_wino_output_transform_f32(M.data_ptr(), cout, b, h, w, d, None, cout, None, None, None, 0, raw.data_ptr(), cout, None, cout, st)))
    t1 = timeit(lambda: check(lib.cp_wino_output_transform_stats_f32(M.data_ptr(), cout, b, h, w, d, None, cout, None, None, None, 0, raw.data_ptr(), cout, None, cout, sums.data_ptr(), st)))
    print("wino_out cout %3d: plain %7.1f us, with stats %7.1f us" % (cout, t0, t1))
