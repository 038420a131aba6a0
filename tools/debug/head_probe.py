#!/usr/bin/env python3
"""The training step's 1x1 head kernels (csrc/head1x1.hip) alone at the training size (bs 32, 448 x 448: 6.4 M pixels x 32 channels = 822 MB):
time and bytes per launch for the record layout the step uses (9 | 27 of 36 floats) and for dense output rows (what the partial-record
writes cost).   python tools/debug/head_probe.py [lib.so]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    os.environ["CASAPOSE_HIP_LIB"] = os.path.abspath(sys.argv[1])
from casapose_amd import _lib
from casapose_amd._lib import check

if len(sys.argv) > 1:   # an older build of the same ABI may lack the newest entry points: bind what it has
    have = C.CDLL(os.environ["CASAPOSE_HIP_LIB"])
    _lib.SYMBOLS[:] = [s for s in _lib.SYMBOLS if hasattr(have, s[0])]
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
B, H, W, c = 32, 448, 448, 32
n = B * H * W
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, c, device=dev, generator=g)
lab = (torch.rand(B, H // 32, W // 32, device=dev, generator=g) * 9).to(torch.uint8).repeat_interleave(32, 1).repeat_interleave(32, 2).reshape(n).contiguous()
rec = torch.zeros(n, 36, device=dev)
dout = torch.randn(n, 40, device=dev, generator=g)
dx = torch.empty(n, c, device=dev)
red = torch.zeros(9 * 64 + 64, dtype=torch.float64, device=dev)


def tables(classes):
    return (torch.rand(classes, c, device=dev, generator=g) + 0.5, torch.randn(classes, c, device=dev, generator=g) * 0.1)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def row(name, us, mb):
    print("%-64s %8.1f us  %7.0f MB  %5.2f TB/s" % (name, us, mb, mb / us))


xb = n * c * 4 / 1e6
for cout, off, classes in ((9, 0, 1), (27, 9, 9)):
    sc, sh = tables(classes)
    w = torch.randn(c, cout, device=dev, generator=g) * 0.2
    lp = lab.data_ptr() if classes > 1 else None
    dense = torch.zeros(n, cout, device=dev)
    padded = torch.zeros(n, 32, device=dev)
    us = timed(lambda: check(lib.cp_head1x1_fwd_affine_f32(x.data_ptr(), c, n, sc.data_ptr(), sh.data_ptr(), lp, classes, 2, w.data_ptr(), cout, rec.data_ptr() + 4 * off, 36, st)))
    row("fwd_affine  cout %2d into the 36-float records" % cout, us, xb + n * cout * 4 / 1e6)
    us = timed(lambda: check(lib.cp_head1x1_fwd_affine_f32(x.data_ptr(), c, n, sc.data_ptr(), sh.data_ptr(), lp, classes, 2, w.data_ptr(), cout, dense.data_ptr(), cout, st)))
    row("fwd_affine  cout %2d into dense rows of %d floats" % (cout, cout), us, xb + n * cout * 4 / 1e6)
    us = timed(lambda: check(lib.cp_head1x1_fwd_affine_f32(x.data_ptr(), c, n, sc.data_ptr(), sh.data_ptr(), lp, classes, 2, w.data_ptr(), cout, padded.data_ptr(), 32, st)))
    row("fwd_affine  cout %2d into rows of 32 floats" % cout, us, xb + n * cout * 4 / 1e6)
    if hasattr(lib, "cp_head1x1_fwd_affine_record_f32") and off:
        pref = torch.randn(n, off, device=dev, generator=g)
        us = timed(lambda: check(lib.cp_head1x1_fwd_affine_record_f32(x.data_ptr(), c, n, sc.data_ptr(), sh.data_ptr(), lp, classes, 2, w.data_ptr(), cout, pref.data_ptr(), off, off,
                                                                      rec.data_ptr(), 36, st)))
        row("fwd_affine  cout %2d + %d copied floats = whole records" % (cout, off), us, xb + n * (cout + 2 * off) * 4 / 1e6)
    us = timed(lambda: check(lib.cp_head1x1_fwd_f32(x.data_ptr(), c, n, w.data_ptr(), cout, rec.data_ptr() + 4 * off, 36, st)))
    row("fwd (no affine)  cout %2d into the records" % cout, us, xb + n * cout * 4 / 1e6)
    mean, rstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    gamma = torch.ones(classes, c, device=dev)
    doff = 0 if off == 0 else 8   # 16-byte aligned start inside rows of 40
    args = (x.data_ptr(), c, dout.data_ptr() + 4 * doff, 40, 32, n, w.data_ptr(), cout, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), sc.data_ptr(), sh.data_ptr(), lp, classes, 2)
    us = timed(lambda: check(lib.cp_head1x1_bn_bwd_reduce_f32(*args, red.data_ptr(), red.data_ptr() + 8 * classes * 64, st)))
    row("bn_bwd_reduce  cout %2d (x + 32 floats of every gradient row)" % cout, us, xb + n * 128 / 1e6)
    us = timed(lambda: check(lib.cp_head1x1_bn_bwd_apply_f32(*args, red.data_ptr() + 8 * classes * 64, float(n), None, dx.data_ptr(), c, st)))
    row("bn_bwd_apply   cout %2d (x + gradient rows in, dx out)" % cout, us, 2 * xb + n * 128 / 1e6)
    dw = torch.zeros(c, cout, device=dev)
    us = timed(lambda: check(lib.cp_head1x1_wgrad_affine_f32(x.data_ptr(), c, sc.data_ptr(), sh.data_ptr(), lp, classes, 2, dout.data_ptr() + 4 * doff, 40, n, cout, dw.data_ptr(), 0, st)))
    row("wgrad_affine   cout %2d" % cout, us, xb + n * cout * 4 / 1e6)
