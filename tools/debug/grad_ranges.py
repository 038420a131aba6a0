#!/usr/bin/env python3
"""max |dY| of every convolution op's output gradient over a few training steps (the operand a data / weight gradient would convert to fp16
pairs): how far apart the layers sit, i.e. whether ONE power of two on the loss brings them all into fp16's band.
    python tools/debug/grad_ranges.py [steps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from casapose_amd.train_engine import ConvOp, ParamStore, TrainPlan
import casapose_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
b, h, w, k = 8, 448, 448, 9
dev = torch.device("cuda:0")
store = ParamStore(O.init_params(k, 27, seed=1, dtype=np.float32), dev)
plan = TrainPlan(store, k, 27, b, h, w)
g = torch.Generator(device="cpu").manual_seed(0)
img = torch.rand(b, h, w, 3, generator=g).to(dev)
lab = torch.zeros(b, h, w, dtype=torch.uint8)
for c in range(1, k):
    y0, x0 = (37 * c) % (h - 120), (53 * c) % (w - 120)
    lab[:, y0:y0 + 100, x0:x0 + 110] = c
lab = lab.to(dev)
kpts = (torch.rand(b, k - 1, 9, 2, generator=g) * min(h, w)).to(dev)
stream = torch.cuda.current_stream(dev).cuda_stream
plan.refresh_weights(stream)
seen = {}
# simplest: read the gradient tensors after the backward (they are kept: the tape's tensors hold .grad)
for s in range(steps):
    plan.train_step(img, lab, lab, kpts, 1e-3, cond_labels=lab, weights=(1.0, 0.5, 0.015))
    torch.cuda.synchronize()
    row = {}
    for op in plan.ops:
        if isinstance(op, ConvOp) and op.out is not None and getattr(op.out, "grad", None) is not None:
            row[op.layer.name] = float(op.out.grad.abs().max())
    seen[s] = row
names = list(seen[0].keys())
print("%-34s" % "op (max |dY| per step)" + "".join("%11d" % s for s in range(steps)))
for n in names:
    print("%-34s" % n + "".join("%11.3g" % seen[s].get(n, float("nan")) for s in range(steps)))
for s in range(steps):
    v = np.array([x for x in seen[s].values() if x > 0])
    print("step %d: largest %.3g, smallest %.3g, ratio 2^%.1f" % (s, v.max(), v.min(), np.log2(v.max() / v.min())))
