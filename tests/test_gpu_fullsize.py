"""Full-size checks (BASELINE.json configs[1]: bs = 16, 480x640, K = 9, ver_dim = 27) through properties that do not need the
oracle to run at that size -- it would take minutes per image: determinism, batch independence (an image's record does not depend
on its neighbours in the batch or on the batch size, which changes every tile -> block assignment), agreement of the independent
kernel routes for the same layers (Winograd vs direct, fused vs unfused resampling / heads), conditioning on the network's own
arg-max vs the same map supplied as `data_segmentation`, and the voting stages on an analytically exact vector field."""
import numpy as np
import pytest
import torch

import casapose_oracle as O

pytestmark = pytest.mark.gpu

B, H, W, K, V = 16, 480, 640, 9, 27


def build(device, seg_input=False, **kw):
    from casapose_amd.pose_models.tfkeras import Classifiers

    net = Classifiers.get("casapose_c_gcu5")(ver_dim=V, seg_dim=K, input_shape=(H, W, 3), input_segmentation_shape=(H, W, K) if seg_input else None,
                                            weights=None, base_model="resnet18", device=device, **kw)
    net.set_parameters(O.init_params(K, V, seed=1237, dtype=np.float32))
    return net


@pytest.fixture(scope="module")
def images():
    g = torch.Generator().manual_seed(1237)
    return 2 * torch.rand(B, H, W, 3, generator=g) - 1


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))


def test_full_batch_determinism_and_batch_independence(device, images):
    net = build(device)
    img = images.to(device)
    out = net([img], training=False).clone()
    assert out.shape == (B, H, W, K + V) and bool(torch.isfinite(out).all())
    again = net([img], training=False)
    assert torch.equal(out, again), "two runs of the same plan must be bit-identical (no atomics on the inference path)"
    # images 3 and 11 alone, in a batch of 2: same records.  Logits bit-for-bit is not required (Winograd tile padding differs with
    # the batch), fp32 round-off is: 1e-5 of the tensor's range.  The vector field is compared where the label maps agree.
    pair = net([img[[3, 11]]], training=False)
    for q, n in enumerate((3, 11)):
        assert rel(pair[q, ..., :K], out[n, ..., :K]) < 1e-5
        la, lb = pair[q, ..., :K].argmax(-1), out[n, ..., :K].argmax(-1)
        agree = la == lb
        assert float(agree.float().mean()) > 0.9999
        # a flipped label changes the field in its 3x3 / pyramid neighbourhood: compare on pixels whose 17x17 surrounding agrees
        bad = torch.nn.functional.max_pool2d((~agree).float()[None, None], 17, 1, 8)[0, 0] > 0
        d = (pair[q, ..., K:] - out[n, ..., K:]).abs().amax(-1)
        assert float(d[~bad].max()) < 1e-4 * float(out[n, ..., K:].abs().max())


def test_full_size_kernel_routes_agree(device, images):
    """Winograd F(4x4,3x3) vs direct MFMA convolution for the nine deep layers, and the fused (resampling inside the conv loader, heads
    inside the halo kernel) vs the materialised route: same network, same weights, conditioning pinned by a supplied mask."""
    img = images[:4].to(device)
    lab = torch.zeros(4, H, W, dtype=torch.long)
    for n in range(4):
        for o in range(1, K):
            y0, x0 = 20 + 50 * ((o + n) % 8), 30 + 70 * ((3 * o + n) % 8)
            lab[n, y0:y0 + 60 + 5 * o, x0:x0 + 40 + 9 * o] = o
    seg = torch.nn.functional.one_hot(lab, K).float().to(device)
    ref = build(device, seg_input=True)([img, seg], training=False).clone()
    direct = build(device, seg_input=True, use_winograd=False)([img, seg], training=False).clone()
    unfused = build(device, seg_input=True, fuse_upsample=False, fuse_heads=False)([img, seg], training=False).clone()
    # F(4x4,3x3) in fp32 carries ~1e-5 relative error per layer; nine layers in sequence
    assert rel(ref, direct) < 5e-4
    assert rel(ref, unfused) < 2e-5


def test_estimated_conditioning_equals_supplied_argmax(device, images):
    """pose_models.py:547-554: the second decoder sees softmax(1e6 * logits) -- a one-hot of the arg-max.  Feeding that map back as
    `data_segmentation` must reproduce the estimated-mask forward exactly (same kernels, same label map)."""
    img = images[:4].to(device)
    est = build(device)([img], training=False).clone()
    seg = torch.nn.functional.one_hot(est[..., :K].argmax(-1), K).float()
    given = build(device, seg_input=True)([img, seg], training=False)
    assert torch.equal(est[..., :K], given[..., :K])
    assert torch.equal(est[..., K:], given[..., K:])


def _exact_field(rng, b, objects, kp):
    """label map of non-overlapping boxes, keypoints, and the unit vector field (dy, dx) pointing at them"""
    lab = np.zeros((b, H, W), np.int64)
    kpts = np.zeros((b, objects, kp, 2), np.float64)  # (y, x)
    for n in range(b):
        for o in range(objects):
            gy, gx = divmod(o, 4)
            y0, x0 = 30 + gy * 220 + int(rng.integers(0, 30)), 20 + gx * 150 + int(rng.integers(0, 20))
            hh, ww = int(rng.integers(60, 150)), int(rng.integers(60, 120))
            lab[n, y0:y0 + hh, x0:x0 + ww] = o + 1
            kpts[n, o, :, 0] = rng.uniform(y0 - 20, y0 + hh + 20, kp)
            kpts[n, o, :, 1] = rng.uniform(x0 - 20, x0 + ww + 20, kp)
    yy, xx = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    dirs = np.zeros((b, H, W, kp, 2), np.float32)
    for n in range(b):
        for o in range(objects):
            m = lab[n] == o + 1
            d = np.stack([kpts[n, o, :, 0][None] - yy[m][:, None], kpts[n, o, :, 1][None] - xx[m][:, None]], -1)
            dirs[n][m] = (d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)).astype(np.float32)
    return lab, kpts, dirs


def test_full_size_ls_voting_recovers_exact_keypoints(device):
    """voting_layers_2d.py:85-122 on an exact field: every pixel's line passes through the keypoint, so the weighted least-squares
    solution is the keypoint for any positive weights; with and without the connected-component filter."""
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted

    rng = np.random.default_rng(11)
    b, objects, kp = 4, K - 1, 9
    lab, kpts, dirs = _exact_field(rng, b, objects, kp)
    seg = torch.nn.functional.one_hot(torch.from_numpy(lab), K).float().to(device) * 10.0  # logits
    direct = torch.from_numpy(dirs.reshape(b, H, W, kp * 2)).to(device)
    conf = torch.from_numpy(rng.standard_normal((b, H, W, kp)).astype(np.float32)).to(device)
    for filt in (False, True):
        got = CoordLSVotingWeighted(name="v", num_classes=K, num_points=kp, filter_estimates=filt)([seg, direct, conf]).cpu().numpy()
        assert np.abs(got - kpts).max() < 2e-2, "keypoints (pixels, y/x) from an exact field"
    # scaling all confidences of one object by a constant must not move its keypoints (weights are per pixel: softplus(conf))
    got2 = CoordLSVotingWeighted(name="v", num_classes=K, num_points=kp)([seg, direct, conf * 0 + 3.0]).cpu().numpy()
    assert np.abs(got2 - kpts).max() < 2e-2


def test_full_size_ransac_voting_recovers_exact_keypoints(device):
    """ransac_voting.py:276-368 on an exact field: every hypothesis is the keypoint, every pixel an inlier, the refinement solves the
    same normal equations -> (x, y) keypoints."""
    from casapose_amd.pose_estimation.ransac_voting import ransac_voting_layer_all_masks

    rng = np.random.default_rng(12)
    b, objects, kp = 2, K - 1, 9
    lab, kpts, dirs = _exact_field(rng, b, objects, kp)
    mask = torch.nn.functional.one_hot(torch.from_numpy(lab), K)[..., 1:].float().to(device)
    vertex = torch.from_numpy(dirs).to(device)
    got = ransac_voting_layer_all_masks(mask, vertex, 512, inlier_thresh=0.99, max_num=30000).cpu().numpy()
    assert got.shape == (b, objects, kp, 2)
    assert np.abs(got[..., ::-1] - kpts).max() < 5e-2  # the RANSAC voter returns (x, y)


def _config2_batch(device, b, h, w):
    """parameters + one synthetic batch at BASELINE.json configs[2]'s shape: ellipse objects on a 3x3 grid, keypoints inside them"""
    rng = np.random.default_rng(21)
    params = O.init_params(K, V, seed=5, dtype=np.float32)
    lab = np.zeros((b, h, w), np.uint8)
    kpts = np.zeros((b, K - 1, 9, 2), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for n in range(b):
        for o in range(K - 1):
            gy, gx = divmod(o, 3)
            cy, cx = 70 + gy * 150 + rng.uniform(-15, 15), 70 + gx * 150 + rng.uniform(-15, 15)
            ry, rx = rng.uniform(25, 60), rng.uniform(25, 60)
            lab[n][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = o + 1
            kpts[n, o, :, 0] = cy + rng.uniform(-ry, ry, 9)
            kpts[n, o, :, 1] = cx + rng.uniform(-rx, rx, 9)
    img = torch.from_numpy(rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)).to(device)
    return params, img, torch.from_numpy(lab).to(device), torch.from_numpy(kpts).to(device)


@pytest.mark.parametrize("mode,tol", [("split", 0.02), ("bf16", 0.02)])
def test_full_size_training_gradient_is_the_directional_derivative(device, monkeypatch, mode, tol):
    """BASELINE.json configs[2] size (bs = 32, 448x448, K = 9): the analytic gradient of the whole step (forward with batch statistics,
    CE + vertex + proxy losses, backward through 28 convolutions incl. the Winograd layers) must predict the change of the loss along
    its own direction: (L(theta + e d) - L(theta - e d)) / 2e = g . d, with d = g / |g|.  In the training plan's default mode (exact
    bf16 splits, fp32-equivalent) and in CASAPOSE_CONV_MODE=bf16 -- configs[2]'s "bf16 convs" -- where the loss itself is the bf16-rounded
    network's (measured ratios 0.9998 and 1.0006)."""
    from casapose_amd.train_engine import ParamStore, TrainPlan

    monkeypatch.setenv("CASAPOSE_CONV_MODE", mode)
    b, h, w = 32, 448, 448
    params, img, labd, kpd = _config2_batch(device, b, h, w)
    store = ParamStore(params, device)
    plan = TrainPlan(store, K, V, b, h, w)
    plan.update_moving = False
    stream = torch.cuda.current_stream(device).cuda_stream
    wts = (1.0, 0.5, 0.015)  # config_8.ini:25-32

    def loss():
        plan.refresh_weights(stream)
        plan.forward(img, cond_labels=labd)
        s = plan.loss_and_grad(labd, labd, kpd, *wts, filter_with_segmentation=False).clone()
        return float(wts[0] * s[0] + wts[1] * s[1] + wts[2] * s[2])

    l0 = loss()
    plan.backward()
    g = store.grad.double().clone()
    gn = float(g.norm())
    assert np.isfinite(l0) and np.isfinite(gn) and gn > 0
    d = (g / gn).float()
    theta0 = store.theta.clone()
    eps = 2e-3
    store.theta.copy_(theta0 + eps * d)
    lp = loss()
    store.theta.copy_(theta0 - eps * d)
    lm = loss()
    store.theta.copy_(theta0)
    fd = (lp - lm) / (2 * eps)
    print("%s mode: directional derivative %.6g vs |g| %.6g (ratio %.4f), loss %.6g" % (mode, fd, gn, fd / gn, l0))
    assert abs(fd - gn) < tol * gn, "directional derivative %.6g vs |g| %.6g (loss %.6g)" % (fd, gn, l0)


# bounds of the bf16 conv mode's GRADIENT at the configs[2] shape against the fp32-equivalent mode on the same parameters and batch (round-3
# verdict: the 799 images/s leg "rides on an ungated gradient").  Measured values are printed by the test; DESIGN.md section 2 quotes them.
# Measured on MI355X (round 4): cosine 0.99643, relative L2 0.0845; per variable median 0.161, 90 % 0.302, worst 0.374 (stage2_unit1_bn1.beta: the
# small per-channel reductions are the cancellation-prone ones); directional ratio 1.0000.
BF16_GRAD_COSINE_MIN = 0.99          # whole flat gradient
BF16_GRAD_REL_L2_MAX = 0.12          # whole flat gradient
BF16_GRAD_VAR_MEDIAN_MAX = 0.22      # per-variable relative L2: median over the 90 variables
BF16_GRAD_VAR_WORST_MAX = 0.50       # ... and the worst variable
BF16_GRAD_DIRECTION_RATIO = (0.98, 1.02)   # fp32-equivalent loss along the bf16 gradient's direction / (g_split . d)


def test_full_size_bf16_gradient_against_the_fp32_equivalent_mode(device, monkeypatch):
    """BASELINE.json configs[2] AS NAMED ("bs=32 bf16 convs", 448x448, K = 9): the gradient the bf16 conv mode hands to Adam, compared with the
    fp32-equivalent mode's on the same parameters and batch -- flat cosine and relative L2, per-variable relative L2 (median and worst of the
    90 variables), and the cross directional derivative: moving along the BF16 gradient's direction d must change the FP32-EQUIVALENT loss
    by g_split . d (finite differences of the exact-split forward), i.e. the bf16 gradient is a descent direction of the loss the other mode
    optimises, with the predicted slope."""
    from casapose_amd.train_engine import ParamStore, TrainPlan

    b, h, w = 32, 448, 448
    params, img, labd, kpd = _config2_batch(device, b, h, w)
    stream = torch.cuda.current_stream(device).cuda_stream
    wts = (1.0, 0.5, 0.015)
    grads, plans = {}, {}
    for mode in ("split", "bf16"):
        monkeypatch.setenv("CASAPOSE_CONV_MODE", mode)
        store = ParamStore(params, device)
        plan = TrainPlan(store, K, V, b, h, w)
        plan.update_moving = False
        plan.refresh_weights(stream)
        plan.forward(img, cond_labels=labd)
        plan.loss_and_grad(labd, labd, kpd, *wts, filter_with_segmentation=False)
        plan.backward()
        torch.cuda.synchronize()
        grads[mode] = store.grad.double().clone()
        if mode == "split":
            plans[mode] = (store, plan)
        else:
            names = {n: store.grad_view(n).double().flatten().clone() for n in store.offsets}
            del plan, store
            torch.cuda.empty_cache()
    store, plan = plans["split"]
    gs, gb = grads["split"], grads["bf16"]
    cos = float(gs @ gb / (gs.norm() * gb.norm()))
    rel = float((gs - gb).norm() / gs.norm())
    per = {}
    for n in store.offsets:
        a = store.grad_view(n).double().flatten()
        per[n] = float((names[n] - a).norm() / max(float(a.norm()), 1e-30))
    vals = sorted(per.values())
    worst = max(per, key=per.get)

    def loss():
        plan.refresh_weights(stream)
        plan.forward(img, cond_labels=labd)
        s_ = plan.loss_and_grad(labd, labd, kpd, *wts, filter_with_segmentation=False).clone()
        return float(wts[0] * s_[0] + wts[1] * s_[1] + wts[2] * s_[2])

    d = (gb / gb.norm()).float()
    theta0 = store.theta.clone()
    eps = 2e-3
    store.theta.copy_(theta0 + eps * d)
    lp = loss()
    store.theta.copy_(theta0 - eps * d)
    lm = loss()
    store.theta.copy_(theta0)
    fd, pred = (lp - lm) / (2 * eps), float(gs @ d.double())
    print("bf16 vs fp32-equivalent gradient at bs 32, 448x448: cosine %.5f, relative L2 %.4f; per variable: median %.4f, 90 %% %.4f, worst %.4f (%s); "
          "fp32-equivalent loss along the bf16 direction: %.6g measured vs %.6g predicted (ratio %.4f)"
          % (cos, rel, vals[len(vals) // 2], vals[int(0.9 * len(vals))], vals[-1], worst, fd, pred, fd / pred))
    assert cos >= BF16_GRAD_COSINE_MIN and rel <= BF16_GRAD_REL_L2_MAX, (cos, rel)
    assert vals[len(vals) // 2] <= BF16_GRAD_VAR_MEDIAN_MAX and vals[-1] <= BF16_GRAD_VAR_WORST_MAX, (vals[len(vals) // 2], worst, vals[-1])
    assert pred > 0 and BF16_GRAD_DIRECTION_RATIO[0] <= fd / pred <= BF16_GRAD_DIRECTION_RATIO[1], (fd, pred)
