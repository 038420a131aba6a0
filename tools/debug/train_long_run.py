#!/usr/bin/env python3
"""Several hundred optimisation steps of the bench's training configuration at a reduced size (bs 8, 224 x 224, fresh random batch every step), once
with forward + backward on fp16 pairs (default) and once on the exact three-way split: loss curves side by side, and what the range machinery did
on the way (loss exponent, exponent moves, GEMMs returned to the exact split).   python tools/debug/train_long_run.py [steps]"""
import os, subprocess, sys, json
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300

if len(sys.argv) > 2:   # worker
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from casapose_amd.train_engine import ParamStore, TrainPlan
    import casapose_oracle as O
    b, h, w, k = 8, 224, 224, 9
    dev = torch.device("cuda:0")
    plan = TrainPlan(ParamStore(O.init_params(k, 27, seed=1, dtype=np.float32), dev), k, 27, b, h, w)
    plan.refresh_weights(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(0)
    lab = torch.zeros(b, h, w, dtype=torch.uint8)
    for c in range(1, k):
        y0, x0 = (37 * c) % (h - 60), (53 * c) % (w - 60)
        lab[:, y0:y0 + 50, x0:x0 + 55] = c
    lab = lab.to(dev)
    hist, exps = [], []
    for s in range(steps):
        img = torch.rand(b, h, w, 3, generator=g).to(dev)
        kpts = (torch.rand(b, k - 1, 9, 2, generator=g) * min(h, w)).to(dev)
        sums = plan.train_step(img, lab, lab, kpts, 1e-3, cond_labels=lab, weights=(1.0, 0.5, 0.015))
        if s % 10 == 0 or s == steps - 1:
            v = sums.cpu().numpy()
            hist.append((s, float(v[0] + 0.5 * v[1] + 0.015 * v[2])))
            exps.append(plan.loss_exp)
    slots = plan._bwd_slots()
    print(json.dumps({"hist": hist, "loss_exp": exps, "moves": len(plan.f16x2_bwd_moves), "demoted": plan.f16x2_demoted, "readings": plan.f16x2_checks,
                      "direct_on": sum(1 for _, f, e in slots if e == "direct" and f["on"]), "direct": sum(1 for _, f, e in slots if e == "direct"),
                      "wino_on": sum(1 for _, f, e in slots if e != "direct" and f["e"] is not None), "wino": sum(1 for _, f, e in slots if e != "direct")}))
    sys.exit(0)

res = {}
for mode in ("f16x2", "split"):
    env = dict(os.environ, CASAPOSE_TRAIN_FWD=mode, CASAPOSE_TRAIN_BWD=mode)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(steps), "worker"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
    res[mode] = json.loads(line[-1])
a, c = res["split"]["hist"], res["f16x2"]["hist"]
print("step   exact split   fp16 pairs   rel. difference   loss exponent")
for (s, la), (_, lc), e in zip(a, c, res["f16x2"]["loss_exp"]):
    print("%4d   %11.5f  %11.5f   %+9.2e        %d" % (s, la, lc, (lc - la) / la, e))
r = res["f16x2"]
print("fp16 pairs: %d / %d direct layers and %d / %d Winograd GEMMs on fp16 pairs at the end; %d range readings, %d exponent moves, returned to the exact split: %s"
      % (r["direct_on"], r["direct"], r["wino_on"], r["wino"], r["readings"], r["moves"], r["demoted"] or "none"))
