"""debug aid: two forwards + filtered vote (tests/test_gpu_forward.py::test_two_forwards_then_filtered_vote_uses_the_right_labels)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import casapose_oracle as O
from casapose_amd import _lib, engine, ops
from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
from casapose_amd.pose_models.tfkeras import Classifiers

dev = torch.device("cuda:0")
b, h, w, k, v = 1, 64, 96, 9, 27
net = Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, base_model="resnet18", device=dev)
net.set_parameters(O.init_params(k, v, seed=1237, dtype=np.float32))
rng = np.random.default_rng(11)
img_a = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
out = net([img_a])
got = out.cpu().numpy()
lab = got[..., :k].argmax(-1)
print("label histogram", np.bincount(lab.ravel(), minlength=k))
lib = _lib.load()
lab0 = torch.from_numpy(lab.astype(np.uint8)).to(dev)
ws = torch.empty(lib.cp_ccl_workspace_bytes(b, h, w, k - 1), dtype=torch.uint8, device=dev)
filt = torch.empty_like(lab0)
_lib.check(lib.cp_ccl_filter_labels(lab0.data_ptr(), b, h, w, k - 1, 50, 1, ws.data_ptr(), filt.data_ptr(), torch.cuda.current_stream().cuda_stream), "ccl")
filt = filt.cpu().numpy()
for o in range(1, k):
    hot = (lab[0] == o).astype(np.int32)
    keep = O.largest_component_filter(hot)
    comp = O.label_components_4(hot > 0)
    sizes = np.sort(np.bincount(comp.ravel())[1:])[::-1][:4]
    print("obj", o, "px", hot.sum(), "top comps", sizes, "oracle keep", int(keep.sum()), "gpu keep", int((filt[0] == o).sum()), "same", bool(((filt[0] == o) == (keep > 0)).all()))
voter = CoordLSVotingWeighted("v", num_classes=k, num_points=9, filter_estimates=True)
s, d, c = torch.split(out, [k, 18, 9], dim=3)
ka = voter([s, d, c]).cpu().numpy()
a = got.astype(np.float64)
ra = O.ls_voting(a[..., :k], a[..., k:k + 18], a[..., k + 18:], filter_estimates=True)
print("max diff per object", np.abs(ka - ra).max(axis=(0, 2, 3)))
rec = out
kp2, sums = ops.ls_vote(rec, 0, k, k + 18, k - 1, 9, labels=torch.from_numpy(filt).to(dev), return_sums=True)
print("vote with gpu-filtered labels vs voter:", np.abs(kp2.cpu().numpy() - ka).max())
keep_all = np.zeros_like(lab)
for o in range(1, k):
    keep_all[0][O.largest_component_filter((lab[0] == o).astype(np.int32)) > 0] = o
kp3, sums3 = ops.ls_vote(rec, 0, k, k + 18, k - 1, 9, labels=torch.from_numpy(keep_all.astype(np.uint8)).to(dev), return_sums=True)
print("vote with oracle-filtered labels vs oracle:", np.abs(kp3.cpu().numpy() - ra).max(axis=(0, 2, 3)))
s5 = sums3.cpu().numpy()[0]
for o in range(k - 1):
    a_, b_, c_ = s5[o, 0, 0], s5[o, 0, 1], s5[o, 0, 2]
    ev = np.linalg.eigvalsh(np.array([[a_, b_], [b_, c_]]))
    print("obj", o + 1, "kp0 system eig", ev, "oracle kp0", ra[0, o, 0], "gpu", kp3.cpu().numpy()[0, o, 0])
