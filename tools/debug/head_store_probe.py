"""timing aid: block 5 / block 10 with their fused heads writing (a) into the merged [pixel][36] record (what the forward does) or (b) into a
compact buffer of their own -- does the partial-record write cost a read-modify-write of the whole record tensor?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from casapose_amd import _lib
from casapose_amd.engine import FusedConv

dev = torch.device("cuda:0")
B, H, W = 16, 480, 640
rng = np.random.default_rng(0)
stream = torch.cuda.current_stream(dev).cuda_stream
hint = {"f32": 7, "split": _lib.TILE_SPLIT3}[os.environ.get("MODE", "f32")]
for name, hc, off in (("b5 + seg head (9 ch)", 9, 0), ("b10 + vertex head (27 ch)", 27, 9)):
    kern = rng.standard_normal((3, 3, 35, 32)).astype(np.float32) / np.sqrt(9 * 35)
    layer = FusedConv(name, kern, 0, 3, 3, 32, [(32, 32), (4, 3)], dev, want_split=True)
    layer.attach_head(rng.standard_normal((1, 1, 32, hc)).astype(np.float32))
    srcs = [dict(data=torch.randn(B, H, W, 32, device=dev), ld=32), dict(data=torch.randn(B, H, W, 4, device=dev), ld=4)]
    sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    rec = torch.zeros(B, H, W, 36, device=dev)
    ldc = (hc + 3) // 4 * 4
    comp = torch.zeros(B, H, W, ldc, device=dev)
    res = {}
    for tag, buf, ld, o in (("record[36]", rec, 36, off), ("compact[%d]" % ldc, comp, ldc, 0)):
        layer.bind(batch=B, in_h=H, in_w=W, pad=1, srcs=srcs, scale=sc, shift=sh, act=_lib.ACT_LEAKY01, head_out=buf, head_out_ld=ld, tile_hint=hint)
        layer.desc.head_out = buf.data_ptr() + 4 * o
        layer.run(stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            layer.run(stream)
        e1.record(); e1.synchronize()
        res[tag] = e0.elapsed_time(e1) / 10
    print(name, "  ".join("%s %.3f ms" % kv for kv in res.items()), flush=True)
