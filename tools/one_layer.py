#!/usr/bin/env python3
"""Run one synthetic convolution repeatedly (for rocprofv3 / tile experiments)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from casapose_amd import ops, _lib
from casapose_amd.engine import FusedConv
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--h", type=int, default=60); ap.add_argument("--w", type=int, default=80)
ap.add_argument("--cin", type=int, default=512); ap.add_argument("--cout", type=int, default=512)
ap.add_argument("--k", type=int, default=3); ap.add_argument("--dil", type=int, default=4)
ap.add_argument("--tile", type=int, default=1); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
x = torch.randn(a.batch, a.h, a.w, a.cin, device=dev)
w = (np.random.default_rng(0).standard_normal((a.k, a.k, a.cin, a.cout)) / np.sqrt(a.k * a.k * a.cin)).astype(np.float32)
layer = FusedConv("bench", w, 0, a.k, a.k, a.cout, [(a.cin, a.cin)], dev)
out = torch.empty(a.batch, a.h, a.w, a.cout, device=dev)
pad = a.dil * (a.k // 2)
layer.bind(batch=a.batch, in_h=a.h, in_w=a.w, dilation=a.dil, pad=pad, srcs=[dict(data=x, ld=a.cin)], out_raw=out, tile_hint=a.tile)
st = torch.cuda.current_stream(dev).cuda_stream
layer.run(st); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps): layer.run(st)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print("tile %d: %.3f ms  %.1f TF/s" % (a.tile, ms, layer.flops / ms / 1e9))
