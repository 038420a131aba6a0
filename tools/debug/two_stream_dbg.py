import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
import casapose_oracle as O
from casapose_amd import engine
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
k, v, b, h, w = 5, 27, 4, 64, 96
for mode in (os.environ.get("M", "f16x2"),):
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, device=dev, conv_mode=mode)
    img = torch.from_numpy(np.random.default_rng(4).uniform(-1, 1, (b, h, w, 3)).astype(np.float32)).to(dev)
    for seed in (3, 8):
        net.set_parameters(O.init_params(k, v, seed=seed, dtype=np.float32))
        engine.TWO_STREAM = False
        one = net([img], training=False).clone()
        engine.TWO_STREAM = True
        two = net([img], training=False).clone()
        torch.cuda.synchronize()
        d = (one - two).abs()
        lab = (one[..., :k].argmax(-1) != two[..., :k].argmax(-1)).float().mean()
        print(mode, "seed", seed, "max diff %.3e of %.3e; seg part %.3e; label mismatch %.2e" % (d.max(), one.abs().max(), d[..., :k].max(), lab),
              "per image", [float(d[i].max()) for i in range(b)])
