#!/usr/bin/env python3
"""Times of the voting-stage kernels on the SURVEY 8(d) synthetic voting inputs (bs 16, 480x640, 8 objects, 9 keypoints)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from casapose_amd import ops
dev = torch.device("cuda:0")
B, H, W, K, KP = 16, 480, 640, 9, 9
g = torch.Generator().manual_seed(1237)
rec = torch.randn(B, H, W, K + 3 * KP, generator=g).to(dev)
lab = torch.randint(0, K, (B, H, W), generator=g, dtype=torch.uint8).to(dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: ops.ls_vote(rec, 0, K, K + 2 * KP, K - 1, KP, labels=lab))
print("ls_vote (labels given, random labels): %.3f ms  %.2f TB/s" % (ms, rec.numel() * 4 / ms / 1e9))
blob = torch.zeros(B, H, W, dtype=torch.uint8)
for o in range(1, K):
    gy, gx = divmod(o - 1, 4)
    blob[:, 40 + gy * 220:40 + gy * 220 + 150, 20 + gx * 150:20 + gx * 150 + 110] = o
blob = blob.to(dev)
ms = t(lambda: ops.ls_vote(rec, 0, K, K + 2 * KP, K - 1, KP, labels=blob))
print("ls_vote (labels given, 8 boxes = 43%% foreground): %.3f ms  %.2f TB/s" % (ms, rec.numel() * 4 / ms / 1e9))
ms = t(lambda: ops.ls_vote(rec, 0, K, K + 2 * KP, K - 1, KP))
print("ls_vote (arg-max inside): %.3f ms  %.2f TB/s" % (ms, rec.numel() * 4 / ms / 1e9))
