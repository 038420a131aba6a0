#!/usr/bin/env python3
"""Does the two-stream forward reproduce the single-stream forward bit for bit?  TRIALS fresh comparisons in one process (a new random batch each);
prints the mismatching trials with the size and place of the difference.  Env: CASAPOSE_INFER_CONV_MODE, CASAPOSE_TWO_STREAM_*."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import casapose_oracle as O
from casapose_amd import engine
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
H, W, K, V, B = 480, 640, 9, 27, int(os.environ.get("BATCH", "16"))
params = O.init_params(K, V, seed=1237, dtype=np.float32)
net = Classifiers.get("casapose_c_gcu5")(ver_dim=V, seg_dim=K, input_shape=(H, W, 3), weights=None, base_model="resnet18", device=dev, seed=1237)
net.set_parameters(params)
bad = 0
for t in range(int(os.environ.get("TRIALS", "8"))):
    img = 2 * torch.rand(B, H, W, 3, device=dev) - 1
    engine.TWO_STREAM = False
    o1 = net([img], training=False).clone()
    engine.TWO_STREAM = True
    for rep in range(int(os.environ.get("REPS", "3"))):
        o2 = net([img], training=False)
        torch.cuda.synchronize()
        if not torch.equal(o1, o2):
            bad += 1
            d = (o1 - o2).abs()
            idx = torch.nonzero(d > 0)
            imgs = sorted(set(idx[:, 0].tolist()))
            chans = sorted(set(idx[:, 3].tolist()))
            ys = idx[:, 1]
            print("trial %d rep %d: %d values differ, max %.3g; images %s; channels %s; rows %d-%d" % (t, rep, idx.shape[0], float(d.max()), imgs, chans[:12], int(ys.min()), int(ys.max())), flush=True)
print("mode %s blocks %d skew %.2f: %d mismatching forwards" % (os.environ.get("CASAPOSE_INFER_CONV_MODE", "f16x2"), engine.TWO_STREAM_BLOCKS, engine.TWO_STREAM_SKEW, bad))
