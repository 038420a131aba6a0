#!/usr/bin/env python3
"""Find what makes the twin of the two-stream forward differ (f16x2 mode): at the first mismatch, run the twin's plan ALONE on its half, compare its
range-guard report and every tensor attribute of its layers with a freshly built net's."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import casapose_oracle as O
from casapose_amd import engine
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
H, W, K, V, B = 480, 640, 9, 27, 16
params = O.init_params(K, V, seed=1237, dtype=np.float32)
net = Classifiers.get("casapose_c_gcu5")(ver_dim=V, seg_dim=K, input_shape=(H, W, 3), weights=None, base_model="resnet18", device=dev, seed=1237)
net.set_parameters(params)
eng = getattr(net, "_net", None)
def find_engine(o):
    for k, v in vars(o).items():
        if isinstance(v, engine.CasaposeNet):
            return v
    return None
eng = eng or find_engine(net)
print("engine:", type(eng).__name__)
def tensors(obj, prefix=""):
    out = {}
    for k, v in vars(obj).items():
        if isinstance(v, torch.Tensor):
            out[prefix + k] = v
        elif isinstance(v, (tuple, list)):
            for i, e in enumerate(v):
                if isinstance(e, torch.Tensor):
                    out["%s%s[%d]" % (prefix, k, i)] = e
        elif isinstance(v, dict):
            for kk, e in v.items():
                if isinstance(e, torch.Tensor):
                    out["%s%s[%s]" % (prefix, k, kk)] = e
    return out
for t in range(10):
    img = 2 * torch.rand(B, H, W, 3, device=dev) - 1
    engine.TWO_STREAM = False
    o1 = net([img], training=False).clone()
    engine.TWO_STREAM = True
    o2 = net([img], training=False)
    torch.cuda.synchronize()
    if torch.equal(o1, o2):
        print("trial %d equal" % t, flush=True)
        continue
    print("trial %d MISMATCH max %.3g" % (t, float((o1 - o2).abs().max())), flush=True)
    tw = eng._twin
    pl = tw.plan(B // 2, H, W)
    engine.TWO_STREAM = False
    a = pl.run(img[B // 2:].contiguous()).clone()
    torch.cuda.synchronize()
    print("  twin plan alone vs single-stream half: equal %s (max %.3g)" % (torch.equal(a, o1[B // 2:]), float((a - o1[B // 2:]).abs().max())))
    pp = eng.plan(B // 2, H, W)
    b = pp.run(img[B // 2:].contiguous()).clone()
    torch.cuda.synchronize()
    print("  primary bs-8 plan on the same half: equal %s (max %.3g)" % (torch.equal(b, o1[B // 2:]), float((b - o1[B // 2:]).abs().max())))
    print("  twin report differs from primary's:", {k: (v, pp.f16x2_report.get(k)) for k, v in pl.f16x2_report.items() if pp.f16x2_report.get(k, (None, None))[1] != v[1]})
    print("  twin fallback:", tw.f16x2_fallback, " primary fallback:", eng.f16x2_fallback)
    # layer tensors: twin vs primary (same parameters -> identical packed weights / tables)
    LA = dict(eng.layers_by_name); LA.update({"wino:" + k: v for k, v in eng.wino_by_name.items()})
    LB = dict(tw.layers_by_name); LB.update({"wino:" + k: v for k, v in tw.wino_by_name.items()})
    names = sorted(set(LA) & set(LB))
    print("  layers compared:", len(names))
    for n in names:
        ta, tb = tensors(LA[n]), tensors(LB[n])
        for k in sorted(set(ta) & set(tb)):
            if ta[k].shape == tb[k].shape and ta[k].dtype == tb[k].dtype and k not in ("out",) and not torch.equal(ta[k], tb[k]):
                d = (ta[k].float() - tb[k].float()).abs()
                print("   layer %s tensor %s differs: %d of %d elements, max %.3g" % (n, k, int((d > 0).sum()), d.numel(), float(d.max())))
        for k in ("head_in_scale", "v_scale", "c_scale", "split_mode", "stem_split"):
            va, vb = getattr(LA[n], k, None), getattr(LB[n], k, None)
            if va != vb:
                print("   layer %s attribute %s: primary %r twin %r" % (n, k, va, vb))
    for k in sorted(set(eng.device_tables) & set(tw.device_tables)):
        for i in range(2):
            if not torch.equal(eng.device_tables[k][i], tw.device_tables[k][i]):
                print("   device table %s[%d] differs" % (k, i))
    break
