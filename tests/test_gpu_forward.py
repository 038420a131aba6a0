"""GPU parity of the whole casapose_c_gcu5 forward and of the LS voter against the fp64
oracle (small images so the oracle runs in seconds), through the reference's own
construction API (`Classifiers.get(name)(...)`, `CoordLSVotingWeighted(...)([...])`)."""
import numpy as np
import pytest
import torch

import casapose_oracle as O

pytestmark = pytest.mark.gpu


def build(device, seg_dim, ver_dim, h, w, seg_input=False, **kw):
    from casapose_amd.pose_models.tfkeras import Classifiers

    ctor = Classifiers.get("casapose_c_gcu5")
    net = ctor(ver_dim=ver_dim, seg_dim=seg_dim, input_shape=(h, w, 3),
               input_segmentation_shape=(h, w, seg_dim) if seg_input else None, weights=None, base_model="resnet18",
               device=device, **kw)
    params = O.init_params(seg_dim, ver_dim, seed=1237, dtype=np.float32)
    net.set_parameters(params)
    return net, {k: v.astype(np.float64) for k, v in params.items()}


def rel_err(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-9)


@pytest.mark.parametrize("fuse", [True, False])
def test_forward_with_given_mask(device, fuse):
    """Decoder 2 conditioned on a supplied one-hot mask (training default, config_8.ini:71;
    pose_models.py:550-554): every output value is compared."""
    b, h, w, k, v = 2, 64, 96, 5, 27
    net, p64 = build(device, k, v, h, w, seg_input=True, fuse_upsample=fuse, fuse_heads=fuse)
    rng = np.random.default_rng(3)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    lab = np.zeros((b, h, w), np.int64)
    lab[:, 8:40, 10:50] = 1
    lab[:, 30:60, 40:90] = 2
    lab[0, 5:20, 60:80] = 3
    lab[1, 44:62, 4:30] = 4
    seg = O.onehot_from_labels(lab, k, np.float32)
    ref = O.casapose_c_gcu5(p64, img.astype(np.float64), seg_input=seg.astype(np.float64))
    out = net([img, seg], training=False)
    got = out.cpu().numpy().astype(np.float64)
    assert got.shape == (b, h, w, k + v)
    # fp32 end to end through 28 conv layers vs fp64: relative 1e-3 of the tensor's range
    assert rel_err(got[..., :k], ref[..., :k]) < 1e-3
    assert rel_err(got[..., k:], ref[..., k:]) < 1e-3


def test_forward_with_estimated_mask(device):
    """Inference path (README.md:74-80): decoder 2 is conditioned on the arg-max of the network's
    own logits.  Logits are compared everywhere.  Label maps may differ only at near-ties of the oracle's
    logits (ties are undefined in the reference), and the vector field is compared at EVERY pixel
    against the oracle's decoder 2 conditioned on the one-hot of the GPU's own label map: with identical
    labels there is nothing left that may differ."""
    b, h, w, k, v = 1, 64, 96, 9, 27
    net, p64 = build(device, k, v, h, w)
    rng = np.random.default_rng(4)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    ref = O.casapose_c_gcu5(p64, img.astype(np.float64))
    got = net([img]).cpu().numpy().astype(np.float64)
    assert rel_err(got[..., :k], ref[..., :k]) < 1e-3
    lab_ref = ref[..., :k].argmax(-1)
    lab_got = got[..., :k].argmax(-1)
    differ = lab_ref != lab_got
    assert differ.mean() <= 1e-3, differ.mean()
    top2 = np.sort(ref[..., :k], -1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    assert (margin[differ] < 1e-3 * np.abs(ref[..., :k]).max()).all()  # only where the oracle itself is at a near-tie
    ref_same_labels = O.casapose_c_gcu5(p64, img.astype(np.float64), seg_input=O.onehot_from_labels(lab_got, k, np.float64))
    assert rel_err(got[..., k:], ref_same_labels[..., k:]) < 1e-3
    if not differ.any():
        assert rel_err(got[..., k:], ref[..., k:]) < 1e-3


def test_two_forwards_then_filtered_vote_uses_the_right_labels(device):
    """Regression (round-1 ADVICE): the voter's component filter reuses the forward's arg-max map.  After
    outA = net(a); outB = net(b) the map cached for outA must still be a's, and an in-place edit of the
    logits must invalidate it."""
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from casapose_amd import engine

    b, h, w, k, v = 1, 64, 96, 9, 27
    net, _ = build(device, k, v, h, w)
    rng = np.random.default_rng(11)
    img_a = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    img_b = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)[:, ::-1].copy()
    out_a = net([img_a])
    out_b = net([img_b])
    got_a, got_b = out_a.cpu().numpy(), out_b.cpu().numpy()
    assert (got_a[..., :k].argmax(-1) != got_b[..., :k].argmax(-1)).any()
    voter = CoordLSVotingWeighted("coords_ls_voting", num_classes=k, num_points=9, filter_estimates=True)

    def vote(out):
        s, d, c = torch.split(out, [k, 18, 9], dim=3)
        return voter([s, d, c]).cpu().numpy()

    def oracle_vote(arr):
        a = arr.astype(np.float64)
        return O.ls_voting(a[..., :k], a[..., k:k + 18], a[..., k + 18:], filter_estimates=True)

    hit = engine.cached_labels(out_a.untyped_storage().data_ptr(), (b, h, w))
    assert hit is not None and (hit.cpu().numpy() == got_a[..., :k].argmax(-1)).all()
    ka, kb = vote(out_a), vote(out_b)
    ra, rb = oracle_vote(got_a), oracle_vote(got_b)

    def close(kp, ref, arr):
        # objects whose kept component has >= 50 pixels: well-posed systems, the 0.05 px gate.  (The reference's top-k rule keeps a LONE
        # component of fewer than 50 pixels -- voting_layers_2d.py:64-76 -- and one or two pixels give a rank-one system whose
        # pseudo-inverse amplifies the last bits of the fp64 sums: those keypoints only have to be finite.)
        lab = arr[0, ..., :k].argmax(-1)
        big = np.array([O.largest_component_filter((lab == o).astype(np.int32)).sum() >= 50 for o in range(1, k)])
        assert big.any() and np.isfinite(kp).all()
        return np.abs(kp[0][big] - ref[0][big]).max() < 0.05 + 1e-4 * np.abs(ref[0][big]).max()

    assert close(ka, ra, got_a) and close(kb, rb, got_b)
    assert not close(ka, rb, got_b)  # and they are different images: a vote on a's output with b's labels would not pass
    # in-place edit of the logits: the cached map is stale and must not be used
    out_a[..., 0] += 100.0  # everything becomes background
    assert engine.cached_labels(out_a.untyped_storage().data_ptr(), (b, h, w)) is None
    assert np.abs(vote(out_a)).max() == 0.0


def test_ls_voting_matches_oracle(device):
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted

    b, h, w, objs = 2, 120, 160, 8
    seg, direct, conf, labels, kps = O.synthetic_voting_inputs(b, h, w, num_obj=objs, seed=7)
    ref = O.ls_voting(seg, direct, conf)
    out = torch.from_numpy(np.concatenate([seg, direct, conf], -1)).to(device)
    s, d, c = torch.split(out, [objs + 1, 18, 9], dim=3)
    got = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=objs + 1, num_points=9)([s, d, c]).cpu().numpy()
    assert got.shape == (b, objs, 9, 2)
    assert np.abs(got - ref).max() < 0.05  # pixels (SURVEY 8d parity gate)
    assert np.abs(got - kps).max() < 2.0   # and it actually finds the keypoints
    # separate (non-view) tensors take the packing path
    got2 = CoordLSVotingWeighted("v", objs + 1)([s.contiguous(), d.contiguous(), c.contiguous()]).cpu().numpy()
    assert np.abs(got2 - ref).max() < 0.05
    # sigmoid_weights=True (voting_layers_2d.py:32-33): another pixel weight, same machinery
    got3 = CoordLSVotingWeighted("v", objs + 1, sigmoid_weights=True)([s, d, c]).cpu().numpy()
    ref3 = O.ls_voting(seg, direct, conf, sigmoid_weights=True)
    assert np.abs(got3 - ref3).max() < 0.05 and np.abs(ref3 - ref).max() > 1e-3


def test_ls_voting_rank_cutoff_is_tensorflows(device):
    """tf.linalg.pinv's default cut-off (10 * 2 * eps = 4.4e-15 of the largest singular value, voting_layers_2d.py:116) against NumPy's 1e-15:
    a two-pixel object whose system is diag(2.6e-15, 1) * 20 must come out rank one (minimum-norm: y = 0); one decade up it is regular.
    The construction and the analytic answers are test_oracle_kat.py::test_ls_voting_rank_cutoff_is_tensorflows_not_numpys."""
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from test_oracle_kat import _two_pixel_field

    for conf_b, regular in ((-30.6, False), (-28.0, True)):
        seg, direct, conf, (ya, xa), (yb, xb) = _two_pixel_field(conf_b)
        rec = torch.from_numpy(np.concatenate([seg, direct, conf], -1)).to(device)
        s, d, c = torch.split(rec, [2, 18, 9], dim=3)
        got = CoordLSVotingWeighted(name="v", num_classes=2, num_points=9)([s, d, c]).cpu().numpy()[0, 0]
        ref = O.ls_voting(seg, direct, conf)[0, 0]
        assert np.abs(got - ref).max() < 1e-3, (conf_b, got, ref)
        assert np.allclose(got[:, 1], xa + 0.5, atol=1e-3)
        assert np.allclose(got[:, 0], yb + 0.5 if regular else 0.0, atol=1e-3)


def test_ls_voting_sums_and_empty_objects(device):
    """fp64 accumulators against the oracle; objects with no pixels give zeros (pinv(0)=0)."""
    from casapose_amd import ops

    b, h, w, objs = 1, 70, 100, 8  # ragged: not multiples of 64 / 16
    seg, direct, conf, labels, _ = O.synthetic_voting_inputs(b, h, w, num_obj=5, seed=9)
    seg = np.concatenate([seg, np.full((b, h, w, objs - 5), -10.0, np.float32)], -1)  # 3 absent objects
    rec = torch.from_numpy(np.concatenate([seg, direct, conf], -1)).to(device)
    kp, sums = ops.ls_vote(rec, 0, objs + 1, objs + 1 + 18, objs, 9, return_sums=True)
    ref = O.ls_voting_sums(seg, direct, conf)
    s = sums.cpu().numpy()
    assert np.abs(s - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())  # fp32 per-pixel terms: ulp-level softplus differences
    assert (kp.cpu().numpy()[:, 5:] == 0).all()


@pytest.mark.parametrize("variant", ["casapose_c", "casapose_c_gu", "casapose_c_gcu3", "casapose_c_gcu4", "casapose_c_gcu4_bilat",
                                     "casapose_c_gcu5_sw5", "casapose_c_gcu4_sw1", "casapose_c_gcu5_sw1", "casapose_c_gcu4_sw2"])
@pytest.mark.parametrize("fuse", [True, False])
def test_registry_variants_forward(device, variant, fuse):
    """CASAPoseConditional1-4, 9 (pose_models.py:14-512,1102-1229): same graph as gcu5 with ordinary convolutions / plain nearest
    upsampling in some decoder-2 blocks; CASAPoseConditional6-8, 10 (:699-1099,1232-1362): the decoders share convolution weights /
    the first convolution's output.  Given mask, every output value compared with the fp64 oracle."""
    from casapose_amd.pose_models.tfkeras import Classifiers

    b, h, w, k, v = 2, 64, 96, 5, 27
    part, _ = O.VARIANTS[variant]
    net = Classifiers.get(variant)(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), input_segmentation_shape=(h, w, k), weights=None,
                                   base_model="resnet18", device=device, fuse_upsample=fuse, fuse_heads=fuse)
    params = O.init_params(k, v, seed=77, dtype=np.float32, partial=part, **O.SHARED.get(variant, {}))
    assert set(params) == set(net.get_parameters()), "variant parameter names"
    net.set_parameters(params)
    rng = np.random.default_rng(5)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    lab = np.zeros((b, h, w), np.int64)
    lab[:, 8:40, 10:50] = 1
    lab[:, 30:60, 40:90] = 2
    lab[0, 5:20, 60:80] = 3
    lab[1, 44:62, 4:30] = 4
    seg = O.onehot_from_labels(lab, k, np.float32)
    ref = O.casapose_c_gcu5({n: a.astype(np.float64) for n, a in params.items()}, img.astype(np.float64), seg_input=seg.astype(np.float64), variant=variant)
    got = net([img, seg], training=False).cpu().numpy().astype(np.float64)
    assert rel_err(got[..., :k], ref[..., :k]) < 1e-3
    assert rel_err(got[..., k:], ref[..., k:]) < 1e-3


def test_custom_decoder_params(device):
    from casapose_amd.pose_models.models.casapose import CASAPOSE_PARAMS, CASAPose, DecoderParams

    params = [DecoderParams(True, i % 2 == 0, i in (1, 3), False, False) for i in range(5)]
    net = CASAPose(params, ver_dim=27, seg_dim=3, input_shape=(32, 32, 3), device=device)
    assert "pv_block_7_conv2d.kernel" in net.get_parameters() and "pv_block_8_prepare_conv2d.weights" in net.get_parameters()
    out = net([np.zeros((1, 32, 32, 3), np.float32)], training=False)
    assert tuple(out.shape) == (1, 32, 32, 30) and bool(torch.isfinite(out).all())
    with pytest.raises(NotImplementedError):
        CASAPose([DecoderParams(True, True, False, True, False)] * 5, ver_dim=27, seg_dim=3, input_shape=(32, 32, 3), device=device)
    assert [tuple(p) for p in CASAPOSE_PARAMS["clade"]][1] == (True, True, True, False, False)


def test_custom_decoder_params_reuse_conv(device):
    """DecoderParams.reuse_conv of the generic builder (casapose.py:178-197,236-261; round-2 verdict "small leaves"): blocks 1/6 and 3/8 share
    one one-input PartialConvolution each (an ordinary SAME convolution on both sides), block 6 normalises block 1's raw convolution output;
    the other decoder-2 blocks keep their own mask-aware convolutions.  Every output value against the fp64 oracle, inference and training."""
    import torch_train_ref as R
    from casapose_amd.pose_models.models.casapose import CASAPose, DecoderParams
    from casapose_amd.train_engine import ParamStore, TrainPlan

    reuse = (True, False, True, False, False)
    dp = [DecoderParams(True, True, i in (1, 2, 3), False, reuse[i]) for i in range(5)]
    part = tuple(p.partial_conv and not p.reuse_conv for p in dp)
    guid = tuple(p.guided_upsampling for p in dp)
    sharing = dict(shared=reuse, reuse_first=True, skips2=True)
    O.VARIANTS["custom_reuse"], O.SHARED["custom_reuse"] = (part, guid), sharing
    try:
        b, h, w, k, v = 2, 64, 96, 4, 27
        net = CASAPose(dp, ver_dim=v, seg_dim=k, input_shape=(h, w, 3), input_segmentation_shape=(h, w, k), device=device)
        params = O.init_params(k, v, seed=78, dtype=np.float32, partial=part, **sharing)
        assert set(params) == set(net.get_parameters()) and "pv_block_1_6_conv2d.weights" in params and "pv_block_3_8_conv2d.weights" in params
        net.set_parameters(params)
        rng = np.random.default_rng(6)
        img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
        lab = np.zeros((b, h, w), np.int64)
        lab[:, 8:40, 10:50], lab[:, 30:60, 40:90], lab[0, 5:20, 60:80] = 1, 2, 3
        seg = O.onehot_from_labels(lab, k, np.float32)
        ref = O.casapose_c_gcu5({n: a.astype(np.float64) for n, a in params.items()}, img.astype(np.float64), seg_input=seg.astype(np.float64), variant="custom_reuse")
        got = net([img, seg], training=False).cpu().numpy().astype(np.float64)
        assert rel_err(got[..., :k], ref[..., :k]) < 1e-3 and rel_err(got[..., k:], ref[..., k:]) < 1e-3
        # training forward (batch statistics) of the same configuration
        plan = TrainPlan(ParamStore(params, device), k, v, b, h, w, partial=part, guided=guid, **sharing)
        plan.refresh_weights(torch.cuda.current_stream(device).cuda_stream)
        out_t = plan.forward(torch.from_numpy(img).to(device), cond_labels=torch.from_numpy(lab.astype(np.uint8)).to(device)).cpu().numpy().astype(np.float64)
        with torch.no_grad():
            ref_t = R.forward_train(R.to_torch(params, requires_grad=False), torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab), partial=part, guided=guid,
                                    **sharing).numpy()
        assert rel_err(out_t, ref_t) < 1e-3
    finally:
        O.VARIANTS.pop("custom_reuse"), O.SHARED.pop("custom_reuse")


def test_pvnet_with_separated_vector_fields_forward(device):
    """the `pvnet` registry entry (models_factory.py:31): per-object vector fields, ver_dim = 2 * points * objects -> a 1x1 head with
    seg_dim + 144 output channels for 8 objects; inference forward vs the oracle (its training step: the next test)."""
    from casapose_amd.pose_models.tfkeras import Classifiers

    b, h, w, k = 1, 32, 48, 9
    v = 2 * 9 * (k - 1)
    params = O.init_params(k, v, seed=33, dtype=np.float32, pvnet=True)
    net = Classifiers.get("pvnet")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, device=device)
    net.set_parameters(params)
    img = np.random.default_rng(4).uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    ref = O.casapose_c_gcu5({n: a.astype(np.float64) for n, a in params.items()}, img.astype(np.float64), variant="pvnet_combined")
    got = net([img], training=False).cpu().numpy().astype(np.float64)
    assert got.shape == (b, h, w, k + v) and rel_err(got, ref) < 1e-3
    assert tuple(net([img], training=True).shape) == (b, h, w, k + v)


@pytest.mark.parametrize("filt", [False, True])
def test_pvnet_separated_vector_fields_training_step(device, filt):
    """`pvnet` TRAINING with separated vector fields (round-2 verdict, missing #5): compute_loss's per-object branch (train_casapose.py:57,97-125)
    -- vertex = sum_o smooth_l1_loss(slice o, target slice o, one-hot o), proxy = sum_o proxy_voting_loss_v2(slice o, keypoints o, one-hot o) --
    value and the gradient of every variable against the fp64 restatement (oracle/loss_functions_ref.compute_loss_separated on
    oracle/torch_train_ref.forward_train(pvnet=True)), on the device's activation branches; then the host API train_step on the model."""
    import loss_functions_ref as LR
    import torch_train_ref as R
    from casapose_amd.train_engine import ParamStore, TrainPlan

    b, h, w, k, kp = 2, 32, 48, 4, 9
    oc = k - 1
    v = 2 * kp * oc
    params = O.init_params(k, v, seed=35, dtype=np.float32, pvnet=True)
    store = ParamStore(params, device)
    plan = TrainPlan(store, k, v, b, h, w, pvnet=True)
    assert plan.GRAD_LD >= k + v and plan.VERT_OFF == k
    plan.refresh_weights(torch.cuda.current_stream(device).cuda_stream)
    rng = np.random.default_rng(6)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    lab = np.zeros((b, h, w), np.uint8)
    lab[:, 4:20, 6:30], lab[:, 14:30, 20:44], lab[0, 2:10, 30:44] = 1, 2, 3
    kpts = rng.uniform(0, h, (b, oc, kp, 2)).astype(np.float32)
    labd = torch.from_numpy(lab).to(device)
    out = plan.forward(torch.from_numpy(img).to(device))
    t_img, t_lab, t_kp = torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64))
    one_hot = torch.from_numpy(np.eye(k)[lab.astype(np.int64)])
    target = LR.get_all_vectorfields(one_hot, t_kp[:, :, None], t_lab[..., None], True)
    wts = (1.0, 0.5, 0.015)

    def reference(pattern, pre=None):
        p = R.to_torch(params)
        o = R.forward_train(p, t_img, t_lab, pvnet=True, act_pattern=pattern, preact_out=pre)
        return p, o, LR.compute_loss_separated(o[..., :k], one_hot, o[..., k:], target, t_kp[:, :, None], filter_vertex_with_segmentation=filt)

    pre = {}
    p64, ref, (ml, vl, pl) = reference(None, pre)
    assert rel_err(out.cpu().numpy().astype(np.float64), ref.detach().numpy()) < 1e-3
    sums = plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(device), *wts, filter_with_segmentation=filt).cpu().numpy()
    for got, want in zip(sums, (ml, vl, pl)):
        assert abs(got - want.item()) < 1e-3 * abs(want.item()), (sums, ml.item(), vl.item(), pl.item())
    plan.backward()
    torch.cuda.synchronize()
    pattern = plan.activation_pattern()
    flips, total, margin = R.kink_report(pattern, pre)
    assert margin < 1e-4 and flips < 1e-4 * total
    if flips:
        p64, _, (ml, vl, pl) = reference(pattern)
    (wts[0] * ml + wts[1] * vl + wts[2] * pl).backward()
    for name in store.offsets:
        g, gr = store.grad_view(name).cpu().numpy().astype(np.float64), p64[name].grad.numpy()
        assert np.linalg.norm(g - gr) / max(np.linalg.norm(gr), 1e-30) < 1e-3, name
    if filt:
        return
    # host API: the `pvnet` registry entry trains through casapose_amd.training.train_step (12 steps reduce the loss)
    from casapose_amd.pose_models.tfkeras import Classifiers
    from casapose_amd.training import Adam, train_step
    from casapose_amd.utils.learning_rate_schedules import LossWeightHandler

    class Opt:
        train_vectors_with_ground_truth = False
        estimate_coords = False

    net = Classifiers.get("pvnet")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, device=device)
    net.set_parameters(params)
    batch = {"img": torch.from_numpy(img), "target_seg": one_hot.float(), "target_vert": torch.from_numpy(kpts)[:, :, None], "filtered_seg": torch.from_numpy(lab)[..., None]}
    lf, optim = LossWeightHandler(*wts, 0.0), Adam(1e-3)
    hist = [train_step(net, batch, lf, optim, Opt())[0] for _ in range(12)]
    assert np.all(np.isfinite(hist)) and hist[-1] < 0.8 * hist[0], hist


def test_pvnet_combined_forward_and_training(device):
    """PVNet baseline (pose_models.py:645-696): decoder 1 + one merged 1x1 head; forward vs the fp64 oracle, and one training
    step's gradients vs the autograd oracle."""
    import torch_train_ref as R
    from casapose_amd.pose_models.tfkeras import Classifiers
    from casapose_amd.train_engine import ParamStore, TrainPlan

    b, h, w, k, v = 2, 32, 48, 4, 27
    params = O.init_params(k, v, seed=31, dtype=np.float32, pvnet=True)
    net = Classifiers.get("pvnet_combined")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, device=device)
    assert set(params) == set(net.get_parameters())
    net.set_parameters(params)
    rng = np.random.default_rng(2)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    ref = O.casapose_c_gcu5({n: a.astype(np.float64) for n, a in params.items()}, img.astype(np.float64), variant="pvnet_combined")
    got = net([img], training=False).cpu().numpy().astype(np.float64)
    assert got.shape == (b, h, w, k + v) and rel_err(got, ref) < 1e-3
    # training step
    store = ParamStore(params, device)
    plan = TrainPlan(store, k, v, b, h, w, pvnet=True)
    plan.refresh_weights(torch.cuda.current_stream(device).cuda_stream)
    lab = np.zeros((b, h, w), np.uint8)
    lab[:, 4:20, 6:30], lab[:, 14:30, 20:44], lab[0, 2:10, 30:44] = 1, 2, 3
    kpts = rng.uniform(0, h, (b, k - 1, 9, 2)).astype(np.float32)
    labd = torch.from_numpy(lab).to(device)
    out = plan.forward(torch.from_numpy(img).to(device))
    p64, pre = R.to_torch(params), {}
    t_img, t_lab, t_kp = torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64))
    ref_t = R.forward_train(p64, t_img, t_lab, pvnet=True, preact_out=pre)
    assert rel_err(out.cpu().numpy().astype(np.float64), ref_t.detach().numpy()) < 1e-3
    sums = plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(device), 1.0, 0.5, 0.015, filter_with_segmentation=False)
    ml, vl, pl = R.losses(ref_t, t_lab, t_kp, k, 9, False)
    plan.backward()
    torch.cuda.synchronize()
    assert abs(float(sums[0]) - ml.item()) < 1e-3 * abs(ml.item())
    # gradients against the oracle on the device's activation branches (tests/test_gpu_train.py::test_train_forward_backward_matches_autograd)
    pattern = plan.activation_pattern()
    flips, total, margin = R.kink_report(pattern, pre)
    assert margin < 1e-4 and flips < 1e-4 * total, (flips, total, margin)
    if flips:
        p64 = R.to_torch(params)
        ml, vl, pl = R.losses(R.forward_train(p64, t_img, t_lab, pvnet=True, act_pattern=pattern), t_lab, t_kp, k, 9, False)
    (ml + 0.5 * vl + 0.015 * pl).backward()
    for name in store.offsets:
        g, gr = store.grad_view(name).cpu().numpy().astype(np.float64), p64[name].grad.numpy()
        assert np.linalg.norm(g - gr) / max(np.linalg.norm(gr), 1e-30) < 1e-3, name


def test_output_lablemap(device):
    """output_lablemap=True (pose_models.py:619-626): [arg-max label as float | vertex] instead of [logits | vertex]."""
    net, _ = build(device, 5, 27, 32, 48)
    from casapose_amd.pose_models.tfkeras import Classifiers

    net2 = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=5, input_shape=(32, 48, 3), weights=None, device=device, output_lablemap=True)
    net2.set_parameters(net.get_parameters())
    img = np.random.default_rng(0).uniform(-1, 1, (1, 32, 48, 3)).astype(np.float32)
    a, b = net([img]).cpu().numpy(), net2([img]).cpu().numpy()
    assert b.shape == (1, 32, 48, 28)
    assert np.array_equal(b[..., 0], a[..., :5].argmax(-1).astype(np.float32)) and np.array_equal(b[..., 1:], a[..., 5:])


def test_load_weights_from_keras_h5(device, tmp_path):
    """net.load_weights(<Keras .h5>, by_name=True, skip_mismatch=True) (test_casapose.py:225-228) through the HDF5 subset reader."""
    from casapose_amd.utils import h5_weights as H
    from test_h5_weights import _keras_like

    net, _ = build(device, 9, 27, 32, 48)
    params = O.init_params(9, 27, seed=99, dtype=np.float32)
    path = str(tmp_path / "result_w_8.h5")
    H.write_h5(path, _keras_like(params))
    net.load_weights(path, by_name=True, skip_mismatch=True)
    got = net.get_parameters()
    assert all(np.array_equal(got[k], params[k]) for k in params)
    # a file for a different class count: mismatching tensors are skipped with a warning, the rest is loaded
    other = O.init_params(5, 27, seed=7, dtype=np.float32)
    path2 = str(tmp_path / "other.h5")
    H.write_h5(path2, _keras_like(other))
    with pytest.warns(UserWarning, match="skipping"):
        net.load_weights(path2)
    got = net.get_parameters()
    assert np.array_equal(got["conv0.kernel"], other["conv0.kernel"]) and np.array_equal(got["pv_block_6_clade.gamma"], params["pv_block_6_clade.gamma"])


@pytest.mark.parametrize("mode,tol", [("f32", 1e-3), ("split", 1e-3), ("f16x2", 1e-3), ("bf16", 3e-2)])
@pytest.mark.parametrize("fuse", [True, False])
def test_forward_in_every_conv_mode(device, mode, tol, fuse):
    """conv_mode="split" (the inference default since round 3): the 3x3 layers off the Winograd path run as exact three-way bf16 splits
    (csrc/conv_hsplit.hip) and the Winograd GEMMs likewise -- the SAME 1e-3 gate as conv_mode="f32", the fp32 MFMA everywhere, which this test
    keeps covered whatever the default is; conv_mode="bf16": operands rounded to bf16 -- the 3e-2 gate of SURVEY 8(d).  fuse=True exercises
    the kernels' fused x2 bilinear / guided sources and 1x1 heads, fuse=False their direct sources."""
    from casapose_amd import _lib

    b, h, w, k, v = 2, 64, 96, 5, 27
    net, p64 = build(device, k, v, h, w, seg_input=True, fuse_upsample=fuse, fuse_heads=fuse, conv_mode=mode)
    rng = np.random.default_rng(3)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    lab = np.zeros((b, h, w), np.int64)
    lab[:, 8:40, 10:50] = 1
    lab[:, 30:60, 40:90] = 2
    lab[0, 5:20, 60:80] = 3
    lab[1, 44:62, 4:30] = 4
    seg = O.onehot_from_labels(lab, k, np.float32)
    ref = O.casapose_c_gcu5(p64, img.astype(np.float64), seg_input=seg.astype(np.float64))
    got = net([img, seg], training=False).cpu().numpy().astype(np.float64)
    assert rel_err(got[..., :k], ref[..., :k]) < tol and rel_err(got[..., k:], ref[..., k:]) < tol
    convs = net._net.plan(b, h, w).convs
    on_pipe = [c.name for c in convs if getattr(c, "split_mode", 0)]
    wino_split = [c.name for c in convs if hasattr(c, "gemm_flops") and getattr(c, "Us", None) is not None]
    if mode == "f32":
        assert not on_pipe and not wino_split and net._net.conv_mode == "f32"
        return
    assert len(on_pipe) >= 10, on_pipe     # stage 1 (4 layers) and decoder blocks 3-5 / 8-10, with fused upsampling / heads or without
    assert all(getattr(c, "split_mode", 0) in (0, {"split": 3, "f16x2": _lib.PLANES_F16X2, "bf16": 1}[mode]) for c in convs)
    deep = [c.name for c in convs if getattr(c, "deep_bf16", False)]
    if mode == "bf16" and deep:   # the deep layers on the direct bf16-operand kernel (csrc/conv_bf16d.hip; CASAPOSE_BF16_DEEP=0: two-plane Winograd)
        assert len(deep) >= 9 and not wino_split, (deep, wino_split)
    else:
        assert len(wino_split) >= 9, wino_split   # the deep layers' Winograd GEMMs on the bf16 pipe too (3 planes / hi + mid)


def test_default_inference_mode_is_the_fp32_level_fp16_split(device):
    from casapose_amd import engine

    import os

    if os.environ.get("CASAPOSE_INFER_CONV_MODE"):
        pytest.skip("the environment overrides the default conv mode for this run")
    from casapose_amd import _lib

    net, _ = build(device, 5, 27, 64, 96)
    assert engine.DEFAULT_INFER_CONV_MODE == "f16x2" and net._net.conv_mode == "f16x2" and net._net.conv_planes == _lib.PLANES_F16X2


def test_bare_resnet18_backbone_taps(device, tmp_path):
    """`Classifiers.get("resnet18")(include_top=False)` -- the registry's bare backbone (models_factory.py:9, resnet.py:374-383; round-2 verdict:
    NotImplementedError): five taps in the reference's output order against the oracle's encoder, the Keras-like surface restricted to the
    encoder's layers, weights through a stand-alone (un-nested) Keras file."""
    from casapose_amd.pose_models.models.resnet import get_backbone
    from casapose_amd.pose_models.tfkeras import Classifiers
    from casapose_amd.utils import h5_weights

    b, h, w = 2, 64, 96
    params = O.init_params(5, 27, seed=51, dtype=np.float32)
    enc = {k: v for k, v in params.items() if h5_weights.is_backbone_layer(k.split(".")[0])}
    net = Classifiers.get("resnet18")(input_shape=(h, w, 3), weights=None, include_top=False, device=device)
    assert set(net.get_parameters()) == set(enc) and {l.name for l in net.layers} == {k.split(".")[0] for k in enc}
    net.set_parameters(enc)
    img = np.random.default_rng(1).uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    taps = net([img], training=False)
    ref = O.resnet18_os8({n: a.astype(np.float64) for n, a in params.items()}, img.astype(np.float64))
    assert [tuple(t.shape) for t in taps] == [(b, 32, 48, 64), (b, 16, 24, 64), (b, 8, 12, 128), (b, 8, 12, 256), (b, 8, 12, 512)]
    for t, r in zip(taps, ref):
        assert rel_err(t.cpu().numpy().astype(np.float64), r) < 1e-3
    # weights round trip through a top-level (not nested) Keras layout; get_backbone() builds the same thing
    path = str(tmp_path / "backbone.h5")
    net.save_weights(path)
    attrs = h5_weights.read_attrs(path)
    assert b"conv0" in attrs["/"]["layer_names"] and b"model" not in attrs["/"]["layer_names"]
    other = get_backbone("resnet18", input_shape=(h, w, 3), weights=path, device=device)
    got = other.get_parameters()
    assert all(np.array_equal(got[k], enc[k]) for k in enc)
    for t, t2 in zip(taps, other(img)):
        assert torch.equal(t, t2)


def test_gemm_route_does_not_survive_a_shape_change(device):
    """Round-2 advisor finding: in the split / bf16 conv modes the 1x1 stride-1 shortcuts run as a bf16-pipe GEMM when rows % 128 == 0; the route
    was remembered across plan rebuilds, so a second shape with rows % 128 != 0 launched the GEMM with the FIRST shape's row count (out of bounds).
    Shape A (applicable) then shape B (not) on one network object, against a conv_mode="f32" network with the same weights."""
    from casapose_amd.pose_models.tfkeras import Classifiers

    k, v = 5, 27
    params = O.init_params(k, v, seed=61, dtype=np.float32)
    nets = {}
    for mode in ("split", "f32"):
        nets[mode] = Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=None, weights=None, device=device, conv_mode=mode)
        nets[mode].set_parameters(params)
    rng = np.random.default_rng(8)
    for b, h, w in ((2, 64, 128), (1, 64, 96), (2, 64, 128)):
        img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
        outs = {m: n([img], training=False).cpu().numpy().astype(np.float64) for m, n in nets.items()}
        convs = nets["split"]._net.plan(b, h, w).convs
        routed = {c.name: c._gemm["rows"] for c in convs if getattr(c, "_gemm", None) is not None}
        for c in convs:   # a route exists exactly where the CURRENT binding's row count allows it, and carries that row count
            if c.name in routed:
                assert routed[c.name] == c.desc.batch * c.desc.out_h * c.desc.out_w and routed[c.name] % 128 == 0, (c.name, routed[c.name])
        rows8 = b * (h // 8) * (w // 8)   # the stage-3 / stage-4 shortcuts run at 1/8 resolution
        assert ("stage4_unit1_sc" in routed) == (rows8 % 128 == 0) and ("stage3_unit1_sc" in routed) == (rows8 % 128 == 0), (routed, rows8)
        assert rel_err(outs["split"][..., :k], outs["f32"][..., :k]) < 1e-4, (b, h, w)


@pytest.mark.parametrize("mode", ["half", "tag"])
def test_two_stream_forward_equals_the_single_stream_forward(device, monkeypatch, mode):
    """CASAPOSE_TWO_STREAM=1 (round 4, opt-in -- measured no faster): the batch as two halves over two HIP streams, either a stream per
    half or a stream per kernel class with events between them, persistent kernels launched with fewer blocks.  The network has no cross-image
    term, so the output must equal the one-stream forward -- bit for bit where both batch sizes take the same kernel routes (bs 16 at 480 x 640:
    `tools/debug/two_stream_bench.py`), to fp32 rounding in general (a 1x1 shortcut whose pixel count is not a multiple of 128 at the half
    batch leaves the bf16-pipe GEMM for the fp32 kernel); the cached label map must serve the filtered voter, and the block count must be
    restored afterwards."""
    from casapose_amd import _lib, engine
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from casapose_amd.pose_models.tfkeras import Classifiers

    k, v, b, h, w = 5, 27, 4, 64, 96
    params = O.init_params(k, v, seed=3, dtype=np.float32)
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), weights=None, device=device)
    net.set_parameters(params)
    img = torch.from_numpy(np.random.default_rng(4).uniform(-1, 1, (b, h, w, 3)).astype(np.float32)).to(device)
    voter = CoordLSVotingWeighted(name="v", num_classes=k, num_points=9, filter_estimates=True)

    def run():
        out = net([img], training=False)
        s, d, c = torch.split(out, [k, 18, 9], dim=3)
        kp = voter([s, d, c])
        torch.cuda.synchronize()
        return out.clone(), kp.clone()

    def same(a, b_):
        # segmentation logits (decoder 1) everywhere; the vector fields are conditioned on the arg-max label map, so ONE pixel whose two top logits
        # tie to the last bits changes them by O(1) inside that pixel's receptive field (seen: 1 of 24576 labels, |difference| 390): they are compared
        # everywhere when the label maps agree, and on all but a small fraction of the pixels otherwise
        lab_a, lab_b = a[..., :k].argmax(-1), b_[..., :k].argmax(-1)
        flips = float((lab_a != lab_b).float().mean())
        seg_ok = float((a[..., :k] - b_[..., :k]).abs().max()) <= 1e-4 * float(a[..., :k].abs().max())
        dv = (a[..., k:] - b_[..., k:]).abs().amax(-1)
        tol = 1e-4 * float(a[..., k:].abs().max())
        vec_ok = float(dv.max()) <= tol if flips == 0.0 else float((dv > tol).float().mean()) <= 2e-2
        return seg_ok and flips <= 1e-3 and vec_ok

    one, kp_one = run()
    lib = _lib.load()
    before = lib.cp_get_persistent_blocks()
    monkeypatch.setattr(engine, "TWO_STREAM", True)
    monkeypatch.setattr(engine, "TWO_STREAM_MODE", mode)
    for _ in range(2):
        two, kp_two = run()
        assert same(one, two)
        assert torch.isfinite(kp_two).all()
        # the label map the filtered voter reads is the two halves' head arg-max maps joined (keypoints themselves are not comparable between
        # runs that differ in the last bits: an untrained network's 2x2 systems are ill-conditioned)
        raw = net([img], training=False)
        cached = engine.cached_labels(raw.untyped_storage().data_ptr(), (b, h, w))
        torch.cuda.synchronize()
        assert cached is not None and torch.equal(cached.long(), raw[..., :k].argmax(-1))
    assert lib.cp_get_persistent_blocks() == before
    # new parameters reach the second half's layer objects too
    params2 = O.init_params(k, v, seed=8, dtype=np.float32)
    net.set_parameters(params2)
    two2, _ = run()
    monkeypatch.setattr(engine, "TWO_STREAM", False)
    one2, _ = run()
    assert same(one2, two2) and not same(one2, one)
