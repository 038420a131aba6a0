"""Committed fixtures (tests/golden/, produced by tests/golden/make_golden.py -- SELF-goldens of
the oracle, see that script's header).  CPU: the oracle still reproduces them.  GPU: the HIP
path reproduces them through the public API."""
import os

import numpy as np
import pytest
import torch

import casapose_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def rel(a, b):
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-9)


# ------------------------------------------------------------------ CPU: oracle vs fixtures ----
def test_oracle_reproduces_forward_fixture():
    g = load("forward_gcu5_k5_32x48.npz")
    p = O.init_params(5, 27, seed=int(g["param_seed"]), dtype=np.float64)
    assert abs(sum(np.abs(a).sum() for a in p.values()) - float(g["param_abs_sum"])) < 1e-6
    seg = O.onehot_from_labels(g["labels"].astype(np.int64), 5)
    out = O.casapose_c_gcu5(p, g["image"].astype(np.float64), seg_input=seg)
    assert rel(out, g["output"]) < 1e-6


def test_oracle_reproduces_layer_and_voting_fixtures():
    g = load("layers_k4_12x16.npz")
    mask = O.onehot_from_labels(g["labels"].astype(np.int64), 4)
    assert rel(O.partial_convolution(g["x"].astype(np.float64), g["weights_ihwo"].astype(np.float64), mask), g["partial_conv"]) < 1e-6
    assert rel(O.guided_upsampling(g["low"].astype(np.float64), O.half_size(mask), mask), g["guided_up"]) < 1e-7
    assert rel(O.guided_bilinear_upsampling(g["low"].astype(np.float64), O.half_size(mask), mask), g["guided_bilinear_up"]) < 1e-6
    v = load("voting_8obj_60x80.npz")
    seg, direct, conf, labels, kps = O.synthetic_voting_inputs(1, 60, 80, num_obj=8, seed=int(v["seed"]))
    assert np.abs(O.ls_voting(seg, direct, conf) - v["ls_keypoints"]).max() < 1e-4
    assert np.abs(v["ls_keypoints"] - v["true_keypoints"]).max() < 2.0
    assert np.abs(v["ransac_keypoints"][..., ::-1] - v["true_keypoints"]).max() < 2.0


# ------------------------------------------------------------------ GPU: HIP vs fixtures -------
@pytest.mark.gpu
def test_hip_forward_matches_fixture(device):
    from casapose_amd.pose_models.tfkeras import Classifiers

    g = load("forward_gcu5_k5_32x48.npz")
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=5, input_shape=(32, 48, 3), input_segmentation_shape=(32, 48, 5),
                                             weights=None, device=device)
    net.set_parameters(O.init_params(5, 27, seed=int(g["param_seed"]), dtype=np.float32))
    seg = O.onehot_from_labels(g["labels"].astype(np.int64), 5, np.float32)
    out = net([g["image"], seg]).cpu().numpy()
    assert rel(out[..., :5], g["output"][..., :5]) < 1e-3
    assert rel(out[..., 5:], g["output"][..., 5:]) < 1e-3


@pytest.mark.gpu
def test_hip_layers_match_fixture(device):
    from casapose_amd import ops

    g = load("layers_k4_12x16.npz")
    lab = torch.from_numpy(g["labels"]).to(device)
    labels, pnorm, sel = ops.label_pyramid(lab)
    x = torch.from_numpy(g["x"]).to(device)
    raw, _ = ops.conv2d_fused([x], g["weights_ihwo"], layout=1, pad=1, tap_label=labels[0], row_scale=pnorm[0])
    assert rel(raw.cpu().numpy(), g["partial_conv"]) < 1e-4
    low = torch.from_numpy(g["low"]).to(device)
    assert rel(ops.guided_upsample_x2(low, sel[0]).cpu().numpy(), g["guided_up"]) < 1e-7
    assert rel(ops.upsample_bilinear_x2(low).cpu().numpy(), g["bilinear_up"]) < 1e-6


@pytest.mark.gpu
def test_hip_ls_voting_matches_fixture(device):
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted

    v = load("voting_8obj_60x80.npz")
    seg, direct, conf, labels, kps = O.synthetic_voting_inputs(1, 60, 80, num_obj=8, seed=int(v["seed"]))
    rec = torch.from_numpy(np.concatenate([seg, direct, conf], -1)).to(device)
    s, d, c = torch.split(rec, [9, 18, 9], dim=3)
    got = CoordLSVotingWeighted("coords_ls_voting", 9)([s, d, c]).cpu().numpy()
    assert np.abs(got - v["ls_keypoints"]).max() < 0.05
