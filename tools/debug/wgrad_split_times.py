"""timing aid: fp32-MFMA weight-gradient kernel vs the bf16-pipe one (exact split / bf16) on the training step's 3x3 shapes (bs 32, 448x448)
   env: B, H, W, ONLY (substring of the shape name), MODES (comma list of f32,split3,bf16), REPS"""
import ctypes as C
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from casapose_amd import _lib
from casapose_amd._lib import ConvDesc, check

dev = torch.device("cuda:0")
lib = _lib.load()
B = int(os.environ.get("B", "32"))
H, W = int(os.environ.get("H", "448")), int(os.environ.get("W", "448"))
REPS = int(os.environ.get("REPS", "10"))
only = os.environ.get("ONLY", "")
modes = os.environ.get("MODES", "f32,split3,bf16").split(",")
shapes = [("stage1 64->64", [64], 64, 4, False), ("stage2 128->128", [128], 128, 8, False), ("b3 128+64->64", [128, 64], 64, 4, False),
          ("b4 64+64->32", [64, 64], 32, 2, False), ("b5 32+img->32", [32, 4], 32, 1, False), ("b6 partial 512->256", [512], 256, 8, True),
          ("b7 partial 256+128->128", [256, 128], 128, 8, True), ("b8 partial 128+64->64", [128, 64], 64, 4, True),
          ("b9 partial 64+64->32", [64, 64], 32, 2, True), ("b10 partial 32+img->32", [32, 4], 32, 1, True)]
stream = torch.cuda.current_stream(dev).cuda_stream
for name, chans, cout, div, partial in shapes:
    if only and only not in name:
        continue
    h, w = H // div, W // div
    xs = [torch.randn(B, h, w, c, device=dev) for c in chans]
    dy = torch.randn(B, h, w, cout, device=dev)
    lab = None
    if partial:
        lab = torch.zeros(B, h, w, dtype=torch.uint8, device=dev)
        lab[:, h // 4: h // 2, w // 4: w // 2] = 1
        lab[:, h // 2:, : w // 3] = 2
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.cout = B, h, w, h, w, cout
    d.kh = d.kw = 3
    d.stride, d.dilation, d.pad = 1, 1, 1
    d.num_sources = len(chans)
    for s, c in enumerate(chans):
        d.src[s].data, d.src[s].channels, d.src[s].ld, d.src[s].mode = xs[s].data_ptr(), c, c, 0
    d.tap_label = lab.data_ptr() if partial else None
    ch = (C.c_int * 2)(*chans, *([0] * (2 - len(chans))))
    ktot = lib.cp_conv_ktot(3, 3, len(chans), ch)
    res, outs = {}, {}
    for tag in modes:
        out = torch.empty(cout, ktot, device=dev)

        def run():
            if tag == "f32":
                check(lib.cp_conv2d_wgrad_f32(C.byref(d), dy.data_ptr(), cout, out.data_ptr(), 0, stream), tag)
            else:
                check(lib.cp_conv2d_wgrad_split(C.byref(d), dy.data_ptr(), cout, out.data_ptr(), 0, 3 if tag == "split3" else 1, stream), tag)
        run()
        torch.cuda.synchronize()
        outs[tag] = out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            run()
        e1.record()
        e1.synchronize()
        res[tag] = e0.elapsed_time(e1) / REPS
    fl = 2.0 * B * h * w * 9 * sum(chans) * cout
    byt = 4.0 * B * h * w * (sum(chans) + cout)
    msg = "%-26s" % name
    for tag in modes:
        msg += "  %s %.3f ms (%5.1f TF-equiv, %.2f TB/s)" % (tag, res[tag], fl / res[tag] / 1e9, byt / res[tag] / 1e9)
    if "f32" in outs:
        for tag in modes:
            if tag != "f32":
                msg += "  maxdiff %s %.1e" % (tag, float((outs[tag] - outs["f32"]).abs().max() / outs["f32"].abs().max()))
    print(msg, flush=True)
