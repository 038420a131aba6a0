#!/usr/bin/env python3
"""Two half-batches on two streams against one batch on one stream (forward only): does an HBM-bound pass of one half overlap the matrix-pipe
kernel of the other when the persistent kernels leave CUs free (CASAPOSE_PERSIST_BLOCKS)?  usage: two_stream_probe.py [offset_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
import casapose_oracle as O
from casapose_amd.pose_models.tfkeras import Classifiers

dev = torch.device("cuda:0")
H, W, K, V = 480, 640, 9, 27
params = O.init_params(K, V, seed=1237, dtype=np.float32)
def make(b):
    n = Classifiers.get("casapose_c_gcu5")(ver_dim=V, seg_dim=K, input_shape=(H, W, 3), weights=None, base_model="resnet18", device=dev, seed=1237)
    n.set_parameters(params)
    return n, (2 * torch.rand(b, H, W, 3, device=dev) - 1)
steps = 30
def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / steps * 1e3
n16, i16 = make(16)
one = timed(lambda: n16([i16], training=False))
del n16; torch.cuda.empty_cache()
na, ia = make(8); nb, ib = make(8)
seq = timed(lambda: (na([ia], training=False), nb([ib], training=False)))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(sa): na([ia], training=False)
    with torch.cuda.stream(sb): nb([ib], training=False)
two = timed(both)
print("persist blocks %s: bs16 one stream %.3f ms | 2 x bs8 sequential %.3f ms | 2 x bs8 on two streams %.3f ms" % (os.environ.get("CASAPOSE_PERSIST_BLOCKS", "256"), one, seq, two))
