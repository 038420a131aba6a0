"""What a pure streaming READ of the LS voter's 708 MB achieves on this box (PyTorch reductions and this library's bn_stats pass), beside the
voter's accumulation kernel: the roofline fraction in bench.py is priced against the 8 TB/s datasheet figure."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from casapose_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
x = torch.randn(16, 480, 640, 36, device=dev)
nbytes = x.numel() * 4
def timed(fn, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
sums = torch.zeros(72, dtype=torch.float64, device=dev)
slot = torch.zeros(4, dtype=torch.int32, device=dev)
for name, fn in (("torch.sum", lambda: x.sum()), ("torch.amax", lambda: x.amax()),
                 ("cp_bn_stats_f32 (36 channels)", lambda: _lib.check(lib.cp_bn_stats_f32(x.data_ptr(), x.numel() // 36, 36, 36, sums.data_ptr(), st))),
                 ("cp_amax_f32 (the new reduction entry point)", lambda: _lib.check(lib.cp_amax_f32(x.data_ptr(), 1, 0, x.numel(), slot.data_ptr(), st))),
                 ("copy (read + write)", lambda: x.clone())):
    t = timed(fn)
    print("%-32s %7.1f us  %5.2f TB/s%s" % (name, t * 1e6, nbytes / t / 1e12, " (x2 bytes moved)" if "copy" in name else ""))
