"""The two CPU oracles agree with each other: the training-mode restatement (oracle/torch_train_ref.py, torch fp64 + autograd) run
with batch statistics equals the inference-mode NumPy restatement (oracle/casapose_oracle.py) once those batch statistics are
written into the moving mean / variance -- for casapose_c_gcu5 and for two registry siblings.  Neither oracle is pinned by
TensorFlow (parity unpinned); this test only guarantees that the GPU's inference and training gates measure against ONE
restatement of the graph, not two."""
import numpy as np
import pytest
import torch

import casapose_oracle as O
import torch_train_ref as R


@pytest.mark.parametrize("variant", ["casapose_c_gcu5", "casapose_c_gcu4_bilat", "casapose_c_gu", "casapose_c_gcu4_sw2"])
def test_training_forward_equals_inference_forward_with_batch_statistics(variant):
    k, v, b, h, w = 4, 27, 2, 32, 32
    partial, guided = O.VARIANTS[variant]
    sharing = O.SHARED.get(variant, O.NOT_SHARED)
    kw = dict(partial=partial, guided=guided, bilinear=O.BILINEAR_GUIDED.get(variant, (False,) * 5), **sharing)
    params = O.init_params(k, v, seed=11, dtype=np.float64, partial=partial, **sharing)
    rng = np.random.default_rng(2)
    img = rng.uniform(-1, 1, (b, h, w, 3))
    lab = np.zeros((b, h, w), np.int64)
    lab[:, 4:20, 6:22] = 1
    lab[0, 14:30, 12:30] = 2
    lab[1, 2:12, 20:31] = 3
    stats = {}
    with torch.no_grad():
        out_t = R.forward_train(R.to_torch(params, requires_grad=False), torch.from_numpy(img), torch.from_numpy(lab), stats_out=stats, **kw).numpy()
    p2 = dict(params)
    for name, (mean, var) in stats.items():
        p2[name + ".moving_mean"], p2[name + ".moving_variance"] = mean.numpy(), var.numpy()
    assert len(stats) >= 29                                              # every normalisation layer of the graph reported
    out_n = O.casapose_c_gcu5(p2, img, seg_input=O.onehot_from_labels(lab, k, np.float64), variant=variant)
    assert out_t.shape == out_n.shape == (b, h, w, k + v)
    assert np.abs(out_t - out_n).max() < 1e-9 * max(1.0, np.abs(out_n).max())


def test_inference_mode_of_the_torch_restatement_equals_the_numpy_oracle():
    """The graph `bench.py`'s cpu_baseline times (forward_train(training=False, labels=None), there in fp32) is the NumPy oracle's
    inference forward with the estimated mask."""
    k, v, b, h, w = 9, 27, 1, 32, 48
    params = O.init_params(k, v, seed=1237, dtype=np.float64)
    img = np.random.default_rng(5).uniform(-1, 1, (b, h, w, 3))
    with torch.no_grad():
        out_t = R.forward_train(R.to_torch(params, requires_grad=False), torch.from_numpy(img), None, training=False).numpy()
    out_n = O.casapose_c_gcu5(params, img)
    assert np.abs(out_t - out_n).max() < 1e-9 * max(1.0, np.abs(out_n).max())
    # and its LS voter equals the NumPy voter on the same record
    seg, direct, conf, labels, _ = O.synthetic_voting_inputs(1, 40, 60, num_obj=8, seed=3)
    got = R.ls_voting(torch.from_numpy(labels.astype(np.int64)), torch.from_numpy(direct.astype(np.float64)), torch.from_numpy(conf.astype(np.float64)), 8).numpy()
    assert np.abs(got - O.ls_voting(seg, direct, conf)).max() < 1e-3
    # the one-pass form bench.py's CPU baseline times (index_add per object instead of a masked reduction per object)
    fast = R.ls_voting_fast(torch.from_numpy(labels.astype(np.int64)), torch.from_numpy(direct), torch.from_numpy(conf), 8).numpy()
    assert np.abs(fast - O.ls_voting(seg, direct, conf)).max() < 1e-3


def test_fast_cpu_inference_path_equals_the_plain_restatement():
    """bench.py's CPU baseline (round 4) times `forward_infer_fast` -- folded normalisation, channels-last, the partial convolution as a 1x1
    convolution to tap planes + masked accumulation.  In fp64 it must equal the plain restatement's inference forward (which the test above
    ties to the NumPy oracle) to re-association error: batch 2, two sizes, and -- same seed and size as above -- the NumPy oracle directly."""
    for b, h, w, seed in ((2, 40, 56, 11), (1, 32, 48, 1237)):
        params = O.init_params(9, 27, seed=seed, dtype=np.float64)
        img = np.random.default_rng(5 if seed == 1237 else 6).uniform(-1, 1, (b, h, w, 3))
        p = R.to_torch(params, requires_grad=False)
        with torch.no_grad():
            out_f = R.forward_infer_fast(R.prepare_inference(p), torch.from_numpy(img)).numpy()
            out_t = R.forward_train(p, torch.from_numpy(img), None, training=False).numpy()
        assert out_f.shape == out_t.shape == (b, h, w, 36)
        assert np.abs(out_f - out_t).max() < 1e-9 * max(1.0, np.abs(out_t).max())
        if seed == 1237:
            out_n = O.casapose_c_gcu5(params, img)
            assert np.abs(out_f - out_n).max() < 1e-9 * max(1.0, np.abs(out_n).max())


def test_functional_loss_restatements_agree_with_the_training_oracle():
    """oracle/loss_functions_ref.py (the reference's FUNCTIONS with their own signatures) against oracle/torch_train_ref.losses (the merged
    compute_loss the training tests use): vertex and proxy terms through smooth_l1_loss / proxy_voting_loss_v2 on get_all_vectorfields'
    target, and the per-object filter values through proxy_voting_dist."""
    import loss_functions_ref as LR

    rng = np.random.default_rng(11)
    b, h, w, k, kp = 2, 24, 32, 4, 9
    lab = np.zeros((b, h, w), np.int64)
    lab[:, 2:12, 3:15], lab[:, 10:22, 16:30], lab[0, 14:22, 2:10] = 1, 2, 3
    one_hot = torch.from_numpy(np.eye(k)[lab])
    kpts = torch.from_numpy(rng.uniform(0, h, (b, k - 1, kp, 2)))
    out = torch.from_numpy(rng.standard_normal((b, h, w, k + 3 * kp)))
    ml, vl, pl = R.losses(out, torch.from_numpy(lab), kpts, k, kp, filter_vertex_with_segmentation=False)
    dirs = out[..., k:k + 2 * kp]
    target = LR.get_all_vectorfields(one_hot, kpts[:, :, None], torch.from_numpy(lab)[..., None], False)
    assert torch.allclose(target, R.target_vector_field(torch.from_numpy(lab), kpts), atol=1e-12)
    v2 = LR.smooth_l1_loss(dirs, target, one_hot[..., 0:1], invert_weights=True)
    p2 = LR.proxy_voting_loss_v2(dirs, kpts[:, :, None], one_hot[..., 1:], one_hot[..., 0:1], invert_weights=True, loss_per_object=False)
    assert abs(v2.item() - vl.item()) < 1e-12 and abs(p2.item() - pl.item()) < 1e-12
    # separated fields: every per-object slice target equals the merged target on that object's pixels
    sep = LR.get_all_vectorfields(one_hot, kpts[:, :, None], torch.from_numpy(lab)[..., None], True)
    for o in range(k - 1):
        m = torch.from_numpy(lab == o + 1)
        assert torch.equal(sep[..., o * 2 * kp:(o + 1) * 2 * kp][m], target[m]) and float(sep[..., o * 2 * kp:(o + 1) * 2 * kp][~m].abs().max()) == 0.0
