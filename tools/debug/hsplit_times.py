"""timing aid: fp32-MFMA halo kernel vs the bf16-pipe kernel (exact split / bf16) on the shallow layers' shapes"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from casapose_amd import _lib, ops
from casapose_amd.engine import FusedConv

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "16"))
H, W = int(os.environ.get("H", "480")), int(os.environ.get("W", "640"))
shapes = [("stage1 64->64", [(64, 64)], 64, 4, False), ("b3 128+64->64", [(128, 128), (64, 64)], 64, 4, False), ("b4 64+64->32", [(64, 64), (64, 64)], 32, 2, False),
          ("b5 32+img->32", [(32, 32), (4, 3)], 32, 1, False), ("b8 partial 128+64->64", [(128, 128), (64, 64)], 64, 4, True),
          ("b9 partial 64+64->32", [(64, 64), (64, 64)], 32, 2, True), ("b10 partial 32+img->32", [(32, 32), (4, 3)], 32, 1, True)]
stream = torch.cuda.current_stream(dev).cuda_stream
rng = np.random.default_rng(0)
for name, sources, cout, div, partial in shapes:
    h, w = H // div, W // div
    cin = sum(s[1] for s in sources)
    kern = (rng.standard_normal((cin, 3, 3, cout)) if partial else rng.standard_normal((3, 3, cin, cout))).astype(np.float32) / np.sqrt(9 * cin)
    layer = FusedConv(name, kern, 1 if partial else 0, 3, 3, cout, sources, dev)
    srcs = [dict(data=torch.randn(B, h, w, s[0], device=dev), ld=s[0]) for s in sources]
    out = torch.empty(B, h, w, cout, device=dev)
    lab = pn = None
    if partial:
        lab0 = torch.zeros(B, h, w, dtype=torch.uint8, device=dev)
        lab0[:, h // 4: h // 2, w // 4: w // 2] = 1
        lab0[:, h // 2:, : w // 3] = 2
        labels, pnorm, _ = ops.label_pyramid(lab0)
        lab, pn = labels[0], pnorm[0]
    res = {}
    outs = {}
    for tag, hint in (("f32", 7), ("split3", _lib.TILE_SPLIT3), ("bf16", _lib.TILE_BF16)):
        layer.bind(batch=B, in_h=h, in_w=w, pad=1, srcs=srcs, tap_label=lab, row_scale=pn, out_raw=out, tile_hint=hint)
        layer.run(stream)
        torch.cuda.synchronize()
        outs[tag] = out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            layer.run(stream)
        e1.record()
        e1.synchronize()
        res[tag] = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * h * w * 9 * cin * cout
    byt = 4.0 * B * h * w * (sum(s[0] for s in sources) + cout)
    d3 = float((outs["split3"] - outs["f32"]).abs().max() / outs["f32"].abs().max())
    d1 = float((outs["bf16"] - outs["f32"]).abs().max() / outs["f32"].abs().max())
    print("%-26s f32 %.3f ms (%5.1f TF)  split3 %.3f ms (%5.1f TF-equiv, %.2f TB/s)  bf16 %.3f ms (%.2f TB/s)  maxdiff split %.1e bf16 %.1e" % (
        name, res["f32"], fl / res["f32"] / 1e9, res["split3"], fl / res["split3"] / 1e9, byt / res["split3"] / 1e9, res["bf16"], byt / res["bf16"] / 1e9, d3, d1), flush=True)
