"""Per-op timing of one training step (forward / backward per tape op) on the GPU.

usage: python tools/train_times.py [batch] [h] [w] [seg_dim]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    from casapose_amd.train_engine import BnActOp, ConvOp, ParamStore, TrainPlan
    import casapose_oracle as O  # parameter initialiser only (tool, not the product path)

    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 448
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 448
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 9
    dev = torch.device("cuda:0")
    params = O.init_params(k, 27, seed=1, dtype=np.float32)
    store = ParamStore(params, dev)
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
    wts = (1.0, 0.5, 0.015)
    for _ in range(2):
        plan.train_step(img, lab, lab, kpts, 1e-3, cond_labels=lab, weights=wts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        plan.train_step(img, lab, lab, kpts, 1e-3, cond_labels=lab, weights=wts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("train step: %.2f ms  (%.1f img/s) at bs=%d %dx%d" % (dt * 1e3, b / dt, b, h, w))

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    # per-op forward
    rows = []
    plan.cond_labels = lab
    for op in plan.ops:
        rows.append([op, timed(lambda: op.forward(stream)), 0.0])
    plan.loss_and_grad(lab, lab, kpts, *wts)
    for t in plan.tensors:
        t.has_grad = False
    for row in reversed(rows):
        row[2] = timed(lambda: row[0].backward(stream))
    tf = tb = 0.0
    print("%-38s %9s %9s %9s" % ("op", "fwd ms", "bwd ms", "fwd TF/s"))
    for op, f, bw in rows:
        if isinstance(op, ConvOp):
            d = op.layer.desc
            cin = sum(s[1] for s in op.layer.sources)
            fl = 2.0 * d.batch * d.out_h * d.out_w * d.kh * d.kw * cin * d.cout
            name, tfs = "conv " + op.layer.name, fl / f / 1e9
            ng = sum(1 for e in op.layer.dgrad if e is not None)
            extra = "  bwd %.1f TF/s (wgrad+%d dgrad)" % (fl * (1 + ng) / max(bw, 1e-6) / 1e9, ng)
        elif isinstance(op, BnActOp):
            name, tfs, extra = "bn   " + op.name, 0.0, "  %.0f MB act" % (op.x.data.numel() * 4 / 1e6)
        else:
            name, tfs, extra = "fn", 0.0, ""
        tf += f
        tb += bw
        print("%-38s %9.3f %9.3f %9.1f%s" % (name, f, bw, tfs, extra))
    print("sum forward %.2f ms, backward %.2f ms" % (tf, tb))
    kinds = {}
    for op, f, bw in rows:
        kk = type(op).__name__
        a = kinds.setdefault(kk, [0.0, 0.0])
        a[0] += f
        a[1] += bw
    for kk, (f, bw) in kinds.items():
        print("  %-10s fwd %8.2f ms  bwd %8.2f ms" % (kk, f, bw))
    loss_t = timed(lambda: plan.loss_and_grad(lab, lab, kpts, *wts))
    adam_t = timed(lambda: store.adam_step(1e-3, stream))
    ref_t = timed(lambda: plan.refresh_weights(stream))
    print("loss %.3f ms, adam %.3f ms, weight refresh %.3f ms" % (loss_t, adam_t, ref_t))


if __name__ == "__main__":
    main()
