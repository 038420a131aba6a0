import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from casapose_amd.pose_models.tfkeras import Classifiers
from casapose_amd import engine
dev = torch.device("cuda:0")
k, v, b, h, w = 9, 27, 2, 96, 128
net = Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, base_model="resnet18", device=dev, seed=1237, conv_mode="f16x2")
img = (2.0 * torch.rand(b, h, w, 3, generator=torch.Generator().manual_seed(1)) - 1.0).to(dev)
net([img], training=False)
plan = net._net.plan(b, h, w)
print(sorted(plan.f16x2_report.keys()))
for c in plan.convs:
    kind = "W" if isinstance(c, engine.WinoConv) else "F"
    print(kind, c.name, getattr(c, "planes", None), getattr(c, "skip_input", None), c.f16x2_active() if kind == "F" else "", c.name in plan.f16x2_report)
