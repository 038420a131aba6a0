#!/usr/bin/env python3
"""Per Winograd layer of the bs-16 forward plan: input transform(s), GEMM and output (or fused output -> input) transform timed separately (HIP
events, 10 launches each), with the bytes each piece moves by construction (V / M fp32 planes, layer input / output) -- the table behind DESIGN.md
4.1c's "what would another out -> in fusion / a direct kernel return".  CASAPOSE_WINO_MIN_K=512 as an argument-free A/B: python wino_parts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import engine
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
B, H, W = 16, 480, 640
net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=9, input_shape=(H, W, 3), weights=None, device=dev, seed=1)
img = (2 * torch.rand(B, H, W, 3) - 1).to(dev)
net([img]); net([img])
plan = net._net.plan(B, H, W)
st = torch.cuda.current_stream(dev).cuda_stream
def timed(fn, reps=10):
    fn(st); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn(st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-22s %4s %5s %5s | %8s %8s %8s | %8s %8s %8s  (us; MB moved: V, M, layer in + out)" % ("layer", "d", "K", "N", "in", "gemm", "out", "V MB", "M MB", "io MB"))
tot = [0.0, 0.0, 0.0]
for c in plan.convs:
    if not isinstance(c, engine.WinoConv): continue
    ms = c.micro_steps()
    t = {"in": 0.0, "gemm": 0.0, "out": 0.0}
    for i, (tag, fn) in enumerate(ms):
        key = "gemm" if tag == "M" else ("in" if i == 0 and len(ms) == 3 else "out")
        t[key] += timed(fn)
    vmb, mmb = 36 * c.Tp * c.ktot * 4 / 1e6, 36 * c.Tp * c.cout * 4 / 1e6
    io = (B * c.h * c.w * (sum(s[0] for s in c.sources) + c.cout) * 4) / 1e6
    note = "  (input written by the layer before: fused out -> in)" if c.skip_input else ("  (out -> in fused into the next layer)" if c.fuse_next is not None else "")
    print("%-22s %4d %5d %5d | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f%s" % (c.name, c.dil, c.ktot, c.cout, t["in"], t["gemm"], t["out"], vmb, mmb, io, note))
    for i, k in enumerate(("in", "gemm", "out")): tot[i] += t[k]
print("sum: input transforms %.0f us, GEMMs %.0f us, output (+ fused) transforms %.0f us" % tuple(tot))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): net._net.forward(img)
e1.record(); e1.synchronize()
print("whole forward: %.3f ms" % (e0.elapsed_time(e1) / 10))
