#!/usr/bin/env python3
"""Segmentation logits of casapose_c_gcu5 at 480 x 640 in the three fp32 arithmetics of the inference plan (f16x2 = default, split, f32) against an fp64
evaluation of the same network on the CPU (oracle/torch_train_ref.forward_infer_fast in double precision), for several parameter seeds and images.
A checker script (it imports oracle/): evidence for DESIGN.md 4.1f, not product path.  usage: accuracy_modes.py [n_seeds] [images_per_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import torch_train_ref as R
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
H, W, seg_dim, ver_dim = 480, 640, 9, 27
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.set_num_threads(min(32, os.cpu_count()))
rows = []
for seed in range(n_seeds):
    rng = np.random.default_rng(100 + seed)
    nets = {m: Classifiers.get("casapose_c_gcu5")(ver_dim=ver_dim, seg_dim=seg_dim, input_shape=(H, W, 3), weights=None, base_model="resnet18", device=dev,
                                                  seed=100 + seed, conv_mode=m) for m in ("f16x2", "split", "f32")}
    params = nets["f32"].get_parameters()
    for k, v in params.items():   # randomised normalisation tables, as in bench.py
        if k.endswith(".gamma") or k.endswith(".moving_variance"):
            params[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith(".beta") or k.endswith(".moving_mean"):
            params[k] = (0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    for n in nets.values():
        n.set_parameters(params)
    q64 = R.prepare_inference(R.to_torch({k: np.asarray(v) for k, v in params.items()}, dtype=torch.float64, requires_grad=False))
    img = (2.0 * torch.rand(n_img, H, W, 3, generator=torch.Generator().manual_seed(seed)) - 1.0)
    with torch.no_grad():
        ref = R.forward_infer_fast(q64, img.double())[..., :seg_dim].numpy()
    for m, net in nets.items():
        got = net([img.to(dev)], training=False)[..., :seg_dim].cpu().numpy().astype(np.float64)
        for i in range(n_img):
            den = np.abs(ref[i]).max()
            e = got[i] - ref[i]
            rows.append((seed, i, m, np.abs(e).max() / den, np.sqrt(np.mean(e ** 2)) / den))
    del nets
    torch.cuda.empty_cache()
print("# max and rms |logit - fp64 logit| / max |fp64 logit| per (parameter seed, image), 480 x 640, K = 9")
print("%-5s %-5s %-6s %10s %10s" % ("seed", "image", "mode", "max", "rms"))
for r in rows:
    print("%-5d %-5d %-6s %10.2e %10.2e" % r)
for m in ("f16x2", "split", "f32"):
    sel = [r for r in rows if r[2] == m]
    print("mean over %d images  %-6s max %.2e  rms %.2e" % (len(sel), m, np.mean([r[3] for r in sel]), np.mean([r[4] for r in sel])))
