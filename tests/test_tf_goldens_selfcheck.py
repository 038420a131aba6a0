"""The consumers in tests/test_tf_goldens.py skip while no TensorFlow-written fixture exists, so a typo in them would stay hidden until the
day the fixtures arrive.  This test writes STAND-IN files of the same format into a temporary directory -- produced by the oracles, NOT by the
reference, so they pin nothing -- points the consumers at them and runs the CPU ones: the day tools/make_tf_goldens.py runs on a
TensorFlow machine, the only thing that can fail is parity itself."""
import os
import sys

import numpy as np
import torch

import casapose_oracle as O
import torch_train_ref as R

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_consumers_run_on_stand_in_files(tmp_path, monkeypatch):
    import make_tf_goldens as M
    import test_tf_goldens as T

    from casapose_amd.pose_estimation import pnp as P
    from casapose_amd.utils import h5_weights as H

    monkeypatch.setattr(T, "GOLD", str(tmp_path))
    # ---- training step ----
    k, h, w = 5, 64, 64
    batch = M.training_batch(np.random.default_rng(7), 2, h, w, k)
    params = O.init_params(k, 27, seed=5, dtype=np.float32)
    p64 = R.to_torch(params)
    lab = torch.from_numpy(batch["labels"])
    out = R.forward_train(p64, torch.from_numpy(batch["img"].astype(np.float64)), lab)
    kpts = torch.from_numpy(batch["target_vert"][:, :, 0].astype(np.float64))
    ml, vl, pl = R.losses(out, lab, kpts, k, 9, True)
    coords = R.ls_voting(lab, out[..., k:k + 18], out[..., k + 18:], k - 1)
    est = torch.argmax(out[..., :k].detach(), -1)
    avail = torch.stack([((est == o).sum((1, 2)) > 50) & ((lab == o).sum((1, 2)) > 50) for o in range(1, k)], 1).double()
    gt_xy = R.project_points(batch["keypoints3d"].reshape(-1, 9, 3).astype(np.float64), batch["cam_mat"][0].astype(np.float64),
                             batch["poses_gt"].reshape(-1, 3, 4).astype(np.float64)).reshape(2, k - 1, 9, 2)
    kl = R.keypoint_reprojection_loss(coords, torch.from_numpy(gt_xy), torch.from_numpy(R.crop_to_image_affine(batch["offsets"].astype(np.float64))), avail,
                                      out[..., k + 18:], lab, 12.5, True)
    total = ml + 0.5 * vl + 0.015 * pl + 0.007 * kl
    total.backward()
    grads = {"grad/" + n: t.grad.numpy() for n, t in p64.items() if t.grad is not None}
    np.savez(tmp_path / "tf_train_k5_64x64.npz", param_seed=5, classes=k, output_training=out.detach().numpy(), coords_yx=coords.detach().numpy(),
             losses=np.array([total.item(), ml.item(), vl.item(), pl.item(), kl.item()]), **{"batch/" + n: a for n, a in batch.items()}, **grads)
    assert len(grads) == 90
    T.test_training_oracle_against_tf()
    # ---- voting with RANSAC ----
    seg, direct, conf, labels, kps = O.synthetic_voting_inputs(1, 60, 80, num_obj=8, seed=31)
    draws = np.random.default_rng(32).integers(0, 2**31 - 1, (20, 1, 8, 128, 9, 2), dtype=np.int64).astype(np.int32)
    pts, rounds = np.zeros((1, 8, 9, 2), np.float32), np.zeros((1, 8), np.int32)
    for o in range(8):
        m = (labels[0] == o + 1).astype(np.float32)
        idx = [draws[r, 0, o].astype(np.int64) % max(int(m.sum()), 1) for r in range(20)]
        pts[0, o], rounds[0, o] = O.ransac_voting_single(m, direct[0].reshape(60, 80, 9, 2), idx)
    np.savez(tmp_path / "tf_voting_8obj_60x80.npz", seg=seg, direct=direct, conf=conf, keypoints_true=kps, ls=O.ls_voting(seg, direct, conf),
             ls_filtered=O.ls_voting(seg, direct, conf, filter_estimates=True), ransac_draws=draws, ransac_keypoints_xy=pts, ransac_rounds=rounds)
    T.test_oracle_voting_against_tf()
    T.test_oracle_ransac_against_tf()
    # ---- Keras HDF5 (stand-in written by our own writer) ----
    H.write_keras_h5(str(tmp_path / "tf_keras_weights_k5.h5"), params)
    np.savez(tmp_path / "tf_keras_weights_k5.npz", param_seed=5, classes=k, keras_reads_our_h5_max_abs_diff=0.0)
    T.test_keras_written_h5_is_read_and_our_h5_is_read_by_keras()
    # ---- PnP (stand-in: our own solver) ----
    rng = np.random.default_rng(41)
    K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], np.float32)
    X = rng.uniform(-60, 60, (4, 9, 3)).astype(np.float32)
    Rm = P.rodrigues(np.array([0.3, -0.2, 0.5]))
    t = np.array([20.0, -10.0, 800.0])
    pix = (X @ Rm.T + t) @ K.T
    x2d = (pix[..., :2] / pix[..., 2:]).astype(np.float32)
    x2d[3] = 0.0
    poses = np.stack([P.pnp(X[i], x2d[i], K, rng=np.random.default_rng(i)) for i in range(4)])
    p6 = np.stack([P.pnp_rvec_t(X[i], x2d[i], K, rng=np.random.default_rng(i)) for i in range(3)])
    up = np.arange(1, 7, dtype=np.float32)
    np.savez(tmp_path / "tf_pnp_cases.npz", points_3d=X, points_2d=x2d, camera=K, poses_cv2=poses, bpnp_pose6=p6, bpnp_upstream=up,
             bpnp_grad_points=np.stack([P.bpnp_backward(up, x2d[i], X[i], K, p6[i]) for i in range(3)]))
    T.test_host_pnp_against_opencv()
