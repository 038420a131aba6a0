"""Bisecting aid: run the smoke-shape training step and dump the gradient of every activation tensor of the tape (after the whole backward),
keyed by op index, into an .npz; `--compare a.npz b.npz` prints, in BACKWARD order, the relative difference of each tensor between two
conv modes so that the first op whose input gradients deviate is visible.
    CASAPOSE_CONV_MODE=f32 python tools/debug/grad_tensor_dump.py --out gpurun_out/gt_f32.npz
    python tools/debug/grad_tensor_dump.py --out gpurun_out/gt_split.npz
    python tools/debug/grad_tensor_dump.py --compare gpurun_out/gt_f32.npz gpurun_out/gt_split.npz"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def dump(a):
    import torch

    import casapose_oracle as O
    from casapose_amd.train_engine import BnActOp, ConvOp, ParamStore, TrainPlan

    dev = torch.device("cuda:0")
    v, tb, th, tw, tk = 27, a.b, a.h, a.w, a.k
    tparams = O.init_params(tk, v, seed=5, dtype=np.float32)
    store = ParamStore(tparams, dev)
    plan = TrainPlan(store, tk, v, tb, th, tw)
    plan.refresh_weights(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(5)
    timg = rng.uniform(-1, 1, (tb, th, tw, 3)).astype(np.float32)
    tlab = np.zeros((tb, th, tw), np.uint8)
    sy, sx = th / 32.0, tw / 32.0
    tlab[:, int(4 * sy):int(20 * sy), int(6 * sx):int(22 * sx)] = 1
    tlab[:, int(14 * sy):int(30 * sy), int(16 * sx):int(30 * sx)] = 2
    if tk > 3:
        tlab[0, int(2 * sy):int(10 * sy), int(20 * sx):int(31 * sx)] = 3
    tkp = rng.uniform(0, th, (tb, tk - 1, 9, 2)).astype(np.float32)
    labd = torch.from_numpy(tlab).to(dev)
    fsnap = {}
    for i, op in enumerate(plan.ops):   # forward-time snapshots: conv outputs, BN statistics and outputs
        def wrapf(i=i, op=op, orig=op.forward):
            def fw(stream):
                orig(stream)
                torch.cuda.synchronize()
                if isinstance(op, ConvOp) and op.out is not None:
                    fsnap["F %03d conv %s out" % (i, op.layer.name)] = op.out.data.clone()
                elif isinstance(op, BnActOp):
                    fsnap["F %03d bn %s x" % (i, op.name)] = op.x.data.clone()
                    fsnap["F %03d bn %s y" % (i, op.name)] = op.y.data.clone()
                    fsnap["F %03d bn %s mean" % (i, op.name)] = op.mean.clone()
                    fsnap["F %03d bn %s rstd" % (i, op.name)] = op.rstd.clone()
                    fsnap["F %03d bn %s scale" % (i, op.name)] = op.scale.clone()
            return fw
        op.forward = wrapf()
    plan.forward(torch.from_numpy(timg).to(dev), cond_labels=labd)
    plan.loss_and_grad(labd, labd, torch.from_numpy(tkp).to(dev), 1.0, 0.5, 0.015, filter_with_segmentation=False)
    out = {}
    snap = {}
    for i, op in enumerate(plan.ops):   # snapshot what every op wrote right after its own backward ("at-time" values)
        def wrap(i=i, op=op, orig=op.backward):
            def bw(stream):
                orig(stream)
                torch.cuda.synchronize()
                if isinstance(op, ConvOp):
                    for s, (t, _) in enumerate(op.srcs):
                        if t.needs_grad and t.has_grad:
                            snap["%03d conv %s dsrc%d %s" % (i, op.layer.name, s, tuple(t.data.shape))] = t.grad.clone()
                    if op.out is not None:
                        snap["%03d conv %s dy" % (i, op.layer.name)] = op.out.grad.clone()
                elif isinstance(op, BnActOp) and op.x.needs_grad and op.x.has_grad:
                    snap["%03d bn %s dx %s" % (i, op.name, tuple(op.x.data.shape))] = op.x.grad.clone()
                    snap["%03d bn %s dy" % (i, op.name)] = op.y.grad.clone()
            return bw
        op.backward = wrap()
    plan.backward()
    torch.cuda.synchronize()
    for k, t in snap.items():
        out["T " + k] = t.cpu().numpy()
    for k, t in fsnap.items():
        out[k] = t.cpu().numpy()
    for i, op in enumerate(plan.ops):
        if isinstance(op, BnActOp):
            out["E %03d bn %s x-at-end" % (i, op.name)] = op.x.data.cpu().numpy()
            out["E %03d bn %s red" % (i, op.name)] = op.red.cpu().numpy()
            out["E %03d bn %s chan" % (i, op.name)] = op.chan.cpu().numpy()
    for i, op in enumerate(plan.ops):
        if isinstance(op, ConvOp):
            for s, (t, _) in enumerate(op.srcs):
                if t.needs_grad and t.has_grad:
                    out["%03d conv %s dsrc%d %s" % (i, op.layer.name, s, tuple(t.data.shape))] = t.grad.cpu().numpy()
            out["%03d conv %s dW" % (i, op.layer.name)] = op.layer.master_grad.cpu().numpy()
            out["%03d conv %s fwd" % (i, op.layer.name)] = (op.out.data if op.out is not None else plan.out).cpu().numpy()
        elif isinstance(op, BnActOp):
            if op.x.needs_grad and op.x.has_grad:
                out["%03d bn %s dx %s" % (i, op.name, tuple(op.x.data.shape))] = op.x.grad.cpu().numpy()
    np.savez(a.out, **out)
    print("wrote", a.out, len(out), "tensors")


def compare(fa, fb):
    A, B = np.load(fa), np.load(fb)
    for k in sorted(A.files, reverse=True):
        x, y = A[k].astype(np.float64), B[k].astype(np.float64)
        n = np.linalg.norm(x)
        print("%-90s rel L2 %.3e   max abs %.3e of %.3e" % (k, np.linalg.norm(x - y) / max(n, 1e-30), np.abs(x - y).max(), np.abs(x).max()))
        if k.startswith("F") and k.endswith(" y"):   # activation outputs: sign disagreements = ReLU / leaky kinks crossed between the two modes
            flip = np.argwhere((x > 0) != (y > 0))
            if len(flip):
                print("      %d sign flips of %d; values there: %s | %s" % (len(flip), x.size, [float("%.3g" % x[tuple(f)]) for f in flip[:6]], [float("%.3g" % y[tuple(f)]) for f in flip[:6]]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--b", type=int, default=2)
    ap.add_argument("--h", type=int, default=32)
    ap.add_argument("--w", type=int, default=32)
    ap.add_argument("--k", type=int, default=4)
    a = ap.parse_args()
    if a.compare:
        compare(*a.compare)
    else:
        dump(a)
