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


# ------------------------------------------------------------------ full BASELINE sizes --------
FULLSIZE = ["fullsize_gcu5_k9_480x640.npz", "fullsize_gcu5_k14_448x448.npz"]


def _fullsize_inputs(g, dtype):
    k, h, w, seed = int(g["classes"]), int(g["height"]), int(g["width"]), int(g["seed"])
    p = O.init_params(k, 27, seed=seed, dtype=dtype)
    img = np.random.default_rng(seed).uniform(-1, 1, (1, h, w, 3)).astype(np.float32)
    assert abs(np.abs(img.astype(np.float64)).sum() - float(g["image_abs_sum"])) < 1e-6 * float(g["image_abs_sum"])
    assert abs(float(sum(np.abs(a.astype(np.float64)).sum() for a in p.values())) - float(g["param_abs_sum"])) < 1e-5 * float(g["param_abs_sum"])
    return k, h, w, p, img


def test_oracle_reproduces_fullsize_fixture():
    """One of the two full-size fixtures (the 480x640 one; ~10 s): the oracle still gives the sampled records and the label map."""
    g = load(FULLSIZE[0])
    k, h, w, p, img = _fullsize_inputs(g, np.float64)
    out = O.casapose_c_gcu5(p, img.astype(np.float64))
    ys, xs = g["sample_y"].astype(np.int64), g["sample_x"].astype(np.int64)
    assert rel(out[0, ys, xs], g["records_estimated"]) < 1e-6
    assert np.array_equal(out[0, ..., :k].argmax(-1), g["labels_estimated"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", FULLSIZE)
def test_hip_forward_matches_fullsize_fixture(device, name):
    """HIP forward at BASELINE.json's sizes (480x640 K=9 = configs[1]; 448x448 K=14 = the 13-object network of configs[4]) against
    the oracle's committed sample of 8192 complete output records:
      * conditioned on the GIVEN label map: every sampled logit and field value within 1e-3 of the tensor's range;
      * estimated mask: logits within 1e-3 everywhere sampled; the label map differs from the oracle's on at most 0.01 % of the pixels
         and only where the oracle's own top-2 margin is a near-tie; the vector field is then compared at EVERY pixel with
        the oracle run live (about 10 s at this size) on decoder 2 conditioned on the GPU's own label map -- identical labels leave
        nothing that may differ, so no neighbourhood has to be excluded."""
    from casapose_amd.pose_models.tfkeras import Classifiers

    g = load(name)
    k, h, w, p, img = _fullsize_inputs(g, np.float32)
    ys, xs = g["sample_y"].astype(np.int64), g["sample_x"].astype(np.int64)
    lr, fr = float(g["logit_range"]), float(g["field_range"])
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=k, input_shape=(h, w, 3), input_segmentation_shape=(h, w, k), weights=None, device=device)
    net.set_parameters(p)
    out = net([img, O.onehot_from_labels(g["labels_given"][None].astype(np.int64), k, np.float32)]).cpu().numpy()[0]
    ref = g["records_given"].astype(np.float64)
    assert np.abs(out[ys, xs][:, :k] - ref[:, :k]).max() < 1e-3 * lr
    assert np.abs(out[ys, xs][:, k:] - ref[:, k:]).max() < 1e-3 * fr
    del net
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=k, input_shape=(h, w, 3), weights=None, device=device)
    net.set_parameters(p)
    out = net([img]).cpu().numpy()[0]
    ref = g["records_estimated"].astype(np.float64)
    assert np.abs(out[ys, xs][:, :k] - ref[:, :k]).max() < 1e-3 * lr
    differ = out[..., :k].argmax(-1) != g["labels_estimated"]
    near_tie = np.unpackbits(g["near_tie"])[:h * w].reshape(h, w).astype(bool)
    assert differ.mean() <= 1e-4 and not (differ & ~near_tie).any(), (differ.mean(), int((differ & ~near_tie).sum()))
    p64 = {n: a.astype(np.float64) for n, a in p.items()}
    live = O.casapose_c_gcu5(p64, img.astype(np.float64), seg_input=O.onehot_from_labels(out[..., :k].argmax(-1)[None].astype(np.int64), k))[0]
    assert np.abs(out[..., :k] - live[..., :k]).max() < 1e-3 * lr
    assert np.abs(out[..., k:] - live[..., k:]).max() < 1e-3 * np.abs(live[..., k:]).max()
    # against the committed records of the oracle's OWN estimated-mask run: where that run's saturated softmax was soft (top-2 margin
    # below ~1e-5: mask values strictly between 0 and 1, undefined behaviour of the reference) the oracle differs from any
    # hard-label evaluation, so a small share of the samples -- those with such a pixel in decoder 2's receptive field -- may disagree
    close = np.abs(out[ys, xs][:, k:] - ref[:, k:]).max(axis=1) < 1e-3 * fr
    assert close.mean() > 0.9, close.mean()
