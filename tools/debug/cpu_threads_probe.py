#!/usr/bin/env python3
"""thread scaling of the CPU restatement's inference graph (oracle/torch_train_ref.forward_infer_fast, fp32) on this host: images/s at bs 1 / 2 / 4"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import casapose_oracle as O, torch_train_ref as R
p = R.to_torch(O.init_params(9, 27, seed=1237, dtype=np.float32), dtype=torch.float32, requires_grad=False)
q = R.prepare_inference(p)
for n in (16, 32, 64, 128):
    if n > os.cpu_count(): break
    torch.set_num_threads(n)
    for bs in (1, 2, 4):
        img = 2.0 * torch.rand(bs, 480, 640, 3) - 1.0
        with torch.no_grad():
            R.forward_infer_fast(q, img)
            t = time.perf_counter(); R.forward_infer_fast(q, img); R.forward_infer_fast(q, img); dt = (time.perf_counter() - t) / 2
        print("threads %3d bs %d: %.2f images/s" % (n, bs, bs / dt), flush=True)
