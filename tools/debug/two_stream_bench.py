#!/usr/bin/env python3
"""single-stream vs two-stream forward (+ voting) at bs 16: equality of the outputs and ms per step over TWO_STREAM_BLOCKS / SKEW settings (env)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import casapose_oracle as O
from casapose_amd import engine
from casapose_amd.pose_models.tfkeras import Classifiers
from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
dev = torch.device("cuda:0")
H, W, K, V, B = 480, 640, 9, 27, int(os.environ.get("BATCH", "16"))
params = O.init_params(K, V, seed=1237, dtype=np.float32)
net = Classifiers.get("casapose_c_gcu5")(ver_dim=V, seg_dim=K, input_shape=(H, W, 3), weights=None, base_model="resnet18", device=dev, seed=1237)
net.set_parameters(params)
img = 2 * torch.rand(B, H, W, 3, device=dev) - 1
voter = CoordLSVotingWeighted(name="v", num_classes=K, num_points=9, filter_estimates=True)
def step():
    out = net([img], training=False)
    s, d, c = torch.split(out, [K, 18, 9], dim=3)
    return out, voter([s, d, c])
def timed(n=30):
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
engine.TWO_STREAM = False
o1, k1 = step(); o1, k1 = o1.clone(), k1.clone()
t1 = timed()
engine.TWO_STREAM = True
o2, k2 = step()
torch.cuda.synchronize()
same = torch.equal(o1, o2)
print("outputs bit-equal: %s (max diff %.3g), keypoints max diff %.3g" % (same, float((o1 - o2).abs().max()), float((k1 - k2).abs().max())))
t2 = timed()
print("blocks %d skew %.2f: single stream %.3f ms (%.1f img/s) | two streams %.3f ms (%.1f img/s)" % (engine.TWO_STREAM_BLOCKS, engine.TWO_STREAM_SKEW, t1, B / t1 * 1e3, t2, B / t2 * 1e3))
