#!/usr/bin/env python3
"""Per-convolution timing table (HIP events on the launch stream) for one forward plan."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from casapose_amd import _lib
from casapose_amd.pose_models.tfkeras import Classifiers

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tile", type=int, default=0, help="force a CP_TILE_* for every conv")
ap.add_argument("--no-fuse", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=9, input_shape=(args.height, args.width, 3), weights=None, device=dev, seed=1, fuse_upsample=not args.no_fuse)
img = (2 * torch.rand(args.batch, args.height, args.width, 3) - 1).to(dev)
net([img])
plan = net._net.plan(args.batch, args.height, args.width)
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
tot_ms = tot_fl = 0
print("%-34s %5s %9s %9s %8s %8s" % ("layer", "tile", "M", "N x K", "ms", "TF/s"))
for c in plan.convs:
    if args.tile: c.desc.tile_hint = args.tile
    t = lib.cp_conv_selected_tile(c.desc)
    c.run(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): c.run(stream)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    d = c.desc
    tot_ms += ms; tot_fl += c.flops
    wino = hasattr(c, "gemm_flops")
    print("%-34s %5s %9d %4dx%-5d %8.3f %8.1f%s" % (c.name, "W" if wino else ("D1" if getattr(c, "deep_bf16", False) else ("P%d" % c.split_mode if getattr(c, "split_mode", 0) else t)), d.batch * d.out_h * d.out_w, d.cout, c.ktot, ms, c.flops / ms / 1e9,
                                                  "   (Winograd: effective rate of the replaced 3x3 conv)" if wino else ""))
print("convs: %.2f ms, %.1f GFLOP, %.1f TF/s" % (tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps): net._net.forward(img)
e1.record(); e1.synchronize()
print("whole forward: %.2f ms" % (e0.elapsed_time(e1) / args.reps))
