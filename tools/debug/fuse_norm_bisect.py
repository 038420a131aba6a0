"""Bisection aid for the fused training normalisation: per-layer statistic tables and activation patterns, fused vs separate passes."""
import os, sys
import numpy as np, torch
sys.path[:0] = [".", "tests", "oracle"]
import test_gpu_train as T
from casapose_amd.train_engine import BnActOp
dev = torch.device("cuda:0")
res, sums, pats = {}, {}, {}
modes = sys.argv[1:] or ["0", "wino_stats"]
for fuse in modes:
    os.environ["CASAPOSE_FUSE_NORM"] = fuse
    params, store, plan, img, lab, kpts = T._setup(dev, 2, 64, 64, 5)
    stream = torch.cuda.current_stream(dev).cuda_stream
    plan.refresh_weights(stream)
    labd = torch.from_numpy(lab).to(dev)
    out = plan.forward(torch.from_numpy(img).to(dev), cond_labels=labd).clone()
    sums[fuse] = {op.name: op.sums.cpu().numpy().copy() for op in plan.ops if isinstance(op, BnActOp)}
    pats[fuse] = plan.activation_pattern()
    plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(dev), 1.0, 0.5, 0.015, filter_with_segmentation=False)
    plan.backward()
    torch.cuda.synchronize()
    res[fuse] = {n: store.grad_view(n).cpu().numpy().copy() for n in store.offsets}
a = modes[0]
for fuse in modes[1:]:
    worst = sorted(((T.rel_l2(res[fuse][n], res[a][n]), n) for n in res[a]), reverse=True)[:6]
    print(fuse, worst)
    for n in sums[a]:
        d = np.abs(sums[fuse][n] - sums[a][n]) / np.maximum(np.abs(sums[a][n]), 1e-30)
        flips = int((pats[fuse][n] != pats[a][n]).sum()) if n in pats[a] else -1
        if d.max() > 1e-9 or flips > 0:
            print("  %-22s sums rel diff max %.2e  activation flips %d of %d" % (n, d.max(), flips, pats[a][n].numel() if n in pats[a] else 0))
