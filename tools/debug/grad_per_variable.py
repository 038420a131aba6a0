"""Per-variable gradient error of one training step against the fp64 autograd oracle (oracle/torch_train_ref.py), at the shape of
__graft_entry__.smoke() by default.  Prints every variable sorted by relative L2 error; used to bisect the round-2 regression the
driver's smoke line showed (2.45e-5 -> 7.49e-3).  Environment switches (CASAPOSE_CONV_MODE, CASAPOSE_WINO_GEMM, CASAPOSE_NO_WINOGRAD,
CASAPOSE_HEAD_CONV) are read by the plan at construction, so each mode is one process:

    python tools/debug/grad_per_variable.py [--b 2 --h 32 --w 32 --k 4] [--top 12]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import casapose_oracle as O
    import torch_train_ref as R
    from casapose_amd.train_engine import ParamStore, TrainPlan

    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=2)
    ap.add_argument("--h", type=int, default=32)
    ap.add_argument("--w", type=int, default=32)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    v = 27
    tb, th, tw, tk = a.b, a.h, a.w, a.k
    tparams = O.init_params(tk, v, seed=a.seed, dtype=np.float32)
    store = ParamStore(tparams, dev)
    plan = TrainPlan(store, tk, v, tb, th, tw)
    plan.refresh_weights(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(a.seed)
    timg = rng.uniform(-1, 1, (tb, th, tw, 3)).astype(np.float32)
    tlab = np.zeros((tb, th, tw), np.uint8)
    sy, sx = th / 32.0, tw / 32.0
    r = lambda y0, y1, x0, x1: (slice(int(y0 * sy), int(y1 * sy)), slice(int(x0 * sx), int(x1 * sx)))  # noqa: E731
    tlab[(slice(None),) + r(4, 20, 6, 22)] = 1
    tlab[(slice(None),) + r(14, 30, 16, 30)] = 2
    if tk > 3:
        tlab[(0,) + r(2, 10, 20, 31)] = 3
    tkp = rng.uniform(0, th, (tb, tk - 1, 9, 2)).astype(np.float32)
    labd = torch.from_numpy(tlab).to(dev)
    out = plan.forward(torch.from_numpy(timg).to(dev), cond_labels=labd).cpu().numpy().astype(np.float64)
    sums = plan.loss_and_grad(labd, labd, torch.from_numpy(tkp).to(dev), 1.0, 0.5, 0.015, filter_with_segmentation=False)
    plan.backward()
    torch.cuda.synchronize()
    from casapose_amd.train_engine import BnActOp

    # the branch every ReLU / leaky pair took in the device forward (its output's sign), by normalisation-layer name
    pattern = {op.name: (op.y.data > 0).cpu() for op in plan.ops if isinstance(op, BnActOp) and op.act != 0}

    def reference(act_pattern):
        p = R.to_torch(tparams)
        pre = {}
        o = R.forward_train(p, torch.from_numpy(timg.astype(np.float64)), torch.from_numpy(tlab.astype(np.int64)), act_pattern=act_pattern, preact_out=pre)
        m_, v_, p_ = R.losses(o, torch.from_numpy(tlab.astype(np.int64)), torch.from_numpy(tkp.astype(np.float64)), tk, 9, False)
        (m_ + 0.5 * v_ + 0.015 * p_).backward()
        return p, o, m_, pre

    p64, ref_out, ml, pre = reference(None)            # the oracle's own branches
    p64g, ref_out_g, _, _ = reference(pattern)         # the oracle evaluated on the branches the device took
    flips, margin = 0, 0.0
    for name, m in pattern.items():
        z = pre[name]
        d = (z > 0) != m
        flips += int(d.sum())
        if d.any():
            margin = max(margin, float(z[d].abs().max()))
    ro = ref_out.detach().numpy()
    print("activation branches: %d of %d differ between the fp32 device forward and the fp64 oracle; largest |pre-activation| among them %.2e; "
          "forward of the oracle on the device's branches moves by %.2e" % (flips, sum(m.numel() for m in pattern.values()), margin,
                                                                          float((ref_out_g - ref_out).detach().abs().max())))
    print("mode: CONV_MODE=%s WINO_GEMM=%s NO_WINOGRAD=%s HEAD_CONV=%s  shape b%d %dx%d K%d" % (
        os.environ.get("CASAPOSE_CONV_MODE", "split"), os.environ.get("CASAPOSE_WINO_GEMM", "split"), os.environ.get("CASAPOSE_NO_WINOGRAD", "0"),
        os.environ.get("CASAPOSE_HEAD_CONV", "stream"), tb, th, tw, tk))
    print("forward: logits %.3e  field %.3e (max abs / range)   mask loss rel err %.3e" % (
        np.abs(out[..., :tk] - ro[..., :tk]).max() / np.abs(ro[..., :tk]).max(), np.abs(out[..., tk:] - ro[..., tk:]).max() / np.abs(ro[..., tk:]).max(),
        abs(float(sums[0]) - ml.item()) / abs(ml.item())))
    for title, ref in (("against the oracle's own branches", p64), ("against the oracle on the DEVICE's branches", p64g)):
        rows = []
        for name in store.offsets:
            g, gr = store.grad_view(name).cpu().numpy().astype(np.float64), ref[name].grad.numpy()
            rows.append((np.linalg.norm(g - gr) / max(np.linalg.norm(gr), 1e-30), name, np.linalg.norm(gr), np.abs(g - gr).max()))
        rows.sort(reverse=True)
        print(title)
        for e, name, n, mx in rows[:a.top]:
            print("  %-42s rel L2 %.3e   |g_ref| %.3e   max abs diff %.3e" % (name, e, n, mx))
        print("worst %.3e (%s)   median %.3e" % (rows[0][0], rows[0][1], rows[len(rows) // 2][0]))


if __name__ == "__main__":
    main()
