"""End-to-end runs of the config-driven drivers (train_casapose.py / test_casapose.py) on the synthetic scene source, and a
known-answer check of the evaluation chain: a network output built from the ground truth must give 100 % ADD recall."""
import csv
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config", "config_8.ini")


def test_evaluation_chain_on_ground_truth_fields(device):
    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset
    from casapose_amd.pose_estimation.pose_evaluation import evaluate_pose_estimates
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from casapose_amd.training import poses_from_coords

    oc, h, w, kp = 8, 448, 448, 9
    ds = SyntheticSceneDataset(oc, (h, w), length=2, seed=11)
    batch = ds.batch(0, 2)
    lab = batch["filtered_seg"][..., 0].numpy()
    kp2 = batch["target_vert"][:, :, 0].numpy()                      # [b,oc,kp,2] (y,x)
    yy, xx = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    dirs = np.zeros((2, h, w, kp, 2), np.float32)
    for n in range(2):
        for o in range(oc):
            m = lab[n] == o + 1
            d = kp2[n, o][None, None] - np.stack([yy, xx], -1)[:, :, None, :]
            d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
            dirs[n][m] = d[m]
    seg = (10.0 * batch["target_seg"]).to(device)
    coords = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=oc + 1, num_points=kp, filter_estimates=True)(
        [seg, torch.from_numpy(dirs.reshape(2, h, w, 2 * kp)).to(device), torch.zeros(2, h, w, kp, device=device)])
    got = coords.cpu().numpy()
    big = batch["pixel_gt_count"].numpy()[:, :, 0, 0] > 200
    assert np.abs(got - kp2)[big].max() < 0.05, "LS voting must return the analytic keypoints (y,x)"
    avail = torch.from_numpy(big.astype(np.float32))
    poses, pts = poses_from_coords(coords, avail, batch)
    stats, _, _ = evaluate_pose_estimates(pts, poses, batch["poses_gt"], batch["target_seg"], batch["keypoints3d"], batch["cam_mat"], batch["diameters"],
                                          evaluation_points=ds.mesh_vertex_array, object_points_3d_count=ds.mesh_vertex_count, min_num=200)
    valid_2d, valid_3d, count_gt = stats[0], stats[1], stats[2]
    assert count_gt.sum() >= 8 and np.all(valid_3d == count_gt) and np.all(valid_2d == count_gt)
    err = np.abs(poses[:, :, 0] - batch["poses_gt"][:, :, 0].numpy())[big]
    assert err[..., :3].max() < 2e-3 and err[..., 3].max() < 1.0     # rotation entries / translation in mm


def test_train_and_test_scripts_end_to_end(device, tmp_path):
    import test_casapose
    import train_casapose

    out = str(tmp_path / "run")
    common = ["-c", CFG, "--outf", out, "--manualseed", "7", "--workers", "0"]
    train_casapose.main(common + ["--data", "synthetic:8", "--datatest", "synthetic:4", "--epochs", "2", "--batchsize", "4", "--imagesize", "128", "160",
                                  "--saveinterval", "1", "--loginterval", "1", "--validationinterval", "2", "--lr_epochs_steps", "1"])
    rows = list(csv.reader(open(out + "/loss_train.csv")))
    assert rows[0][:7] == ["epoch", "batchid", "loss", "mask_loss", "vertex_loss", "proxy_loss", "keypoint_loss"] and len(rows) == 1 + 2 * 2
    vals = np.array([[float(v) for v in r[2:7]] for r in rows[1:]])
    assert np.all(np.isfinite(vals)) and vals[-1, 0] < vals[0, 0]
    summ = list(csv.reader(open(out + "/train_summary.csv")))
    assert len(summ) == 3 and float(summ[1][1]) == 0.001 and float(summ[2][1]) == 0.0005   # lr halves after epoch 1 (lr_decay 0.5)
    tsum = list(csv.reader(open(out + "/test_summary.csv")))
    assert len(tsum[0]) == 7 + 16 and len(tsum) == 3 and len(tsum[2]) == 7 + 16             # pose columns on validation epochs
    assert os.path.exists(out + "/frozen_model/result_w.h5") and os.path.exists(out + "/training_checkpoints/ckpt-3.npz")
    assert os.path.exists(out + "/header.txt")
    res = test_casapose.main(common + ["--datatest", "synthetic:2", "--load_h5_weights", "1", "--imagesize_test", "480", "640", "--write_poses", "1"])
    ev = list(csv.reader(open(out + "/test_summary_eval.csv")))
    assert ev[0][:6] == ["loss", "mask_loss", "vertex_loss", "proxy_loss", "kp_loss", "time"] and len(ev) == 2 and len(ev[1]) == 5 + 9 + 9
    assert len(list(csv.reader(open(out + "/loss_test_eval.csv")))) == 3
    assert np.all(np.isfinite(res["loss"])) and res["valid_3d"].shape == (8,)
    bop = list(csv.reader(open(out + "/poses_out/bop_evaluation.csv")))       # io_utils.py:54-138
    assert bop[0] == ["scene_id", "im_id", "obj_id", "score", "R", "t", "time"] and len(bop) > 1
    assert all(len(r) == 7 and len(r[4].split()) == 9 and len(r[5].split()) == 3 and float(r[3]) in (0.0, 1.0) for r in bop[1:])
    assert os.path.exists(out + "/poses_out/all_poses/poses_init_obj_000001.txt") and os.path.exists(out + "/poses_out/filtered_poses/poses_gt_obj_000001.txt")
    # frozen_model/result_w.h5 is real HDF5 in Keras' layout (train_casapose.py:903); resuming on --net continues the checkpoint numbering
    from casapose_amd.utils import h5_weights

    assert h5_weights.is_hdf5(out + "/frozen_model/result_w.h5")
    attrs = h5_weights.read_attrs(out + "/frozen_model/result_w.h5")
    assert b"model" in attrs["/"]["layer_names"] and b"conv0/kernel:0" in attrs["/model"]["weight_names"]
    before = dict(np.load(out + "/training_checkpoints/ckpt-3.npz"))
    train_casapose.main(common + ["--data", "synthetic:4", "--datatest", "", "--epochs", "1", "--batchsize", "4", "--imagesize", "128", "160",
                                  "--saveinterval", "1", "--lr", "0.0"])
    assert os.path.exists(out + "/training_checkpoints/ckpt-5.npz")            # restored ckpt-3, saved 4 (epoch) and 5 (end of run)
    after = dict(np.load(out + "/training_checkpoints/ckpt-5.npz"))
    trained = [k for k in before if not k.endswith(("moving_mean", "moving_variance"))]
    assert all(np.array_equal(before[k], after[k]) for k in trained)            # lr 0: the restored weights, not a fresh initialisation


def test_test_script_on_an_ndds_folder(device, tmp_path):
    """test_casapose.py reading the reference's on-disk dataset format (exported synthetic scene): reader -> network -> voting ->
    PnP -> ADD statistics.  With random weights the recall is meaningless; the run must complete and write its reports."""
    import test_casapose
    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset
    from casapose_amd.data_handler.vectorfield_dataset import write_ndds_scene

    names = "obj_000001,obj_000005,obj_000006,obj_000008,obj_000009,obj_000010,obj_000011,obj_000012".split(",")
    scene = SyntheticSceneDataset(8, (480, 640), length=2, seed=3)
    write_ndds_scene(str(tmp_path / "data"), str(tmp_path / "models"), scene, 2, names)
    out = str(tmp_path / "run")
    res = test_casapose.main(["-c", CFG, "--outf", out, "--manualseed", "7", "--datatest", str(tmp_path / "data"), "--datameshes", str(tmp_path / "models"),
                              "--net", "", "--pretrained", "0"])
    rows = list(csv.reader(open(out + "/loss_test_eval.csv")))
    assert len(rows) == 3 and np.all(np.isfinite(res["loss"]))


def test_ransac_voting_pose_chain_on_ground_truth_fields(device):
    """estimate_and_evaluate_poses (pose_evaluation.py:11-97; the estimate_coords=0 path of test_casapose.py): RANSAC keypoint voting
    on the arg-max mask -> crop->image transform -> host PnP -> ADD; ground-truth fields of a 13-object scene (config_13.ini) must be
    recovered for every sufficiently visible object."""
    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset
    from casapose_amd.pose_estimation.pose_evaluation import estimate_and_evaluate_poses

    oc, h, w, kp = 13, 448, 448, 9
    ds = SyntheticSceneDataset(oc, (h, w), length=1, seed=4)
    batch = ds.batch(0, 1)
    lab = batch["filtered_seg"][..., 0].numpy()
    kp2 = batch["target_vert"][:, :, 0].numpy()
    yy, xx = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    dirs = np.zeros((1, h, w, kp, 2), np.float32)
    for o in range(oc):
        m = lab[0] == o + 1
        d = kp2[0, o][None, None] - np.stack([yy, xx], -1)[:, :, None, :]
        d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
        dirs[0][m] = d[m]
    seg = (10.0 * batch["target_seg"]).to(device)
    stats, poses, pts = estimate_and_evaluate_poses(seg, batch["target_seg"], torch.from_numpy(dirs.reshape(1, h, w, 2 * kp)).to(device), batch["poses_gt"],
                                                    batch["keypoints3d"], batch["cam_mat"], batch["diameters"], batch["offsets"],
                                                    evaluation_points=ds.mesh_vertex_array, object_points_3d_count=ds.mesh_vertex_count, min_num=200)
    valid_2d, valid_3d, count_gt, false_pos = stats[0], stats[1], stats[2], stats[3]
    assert count_gt.sum() >= 6 and np.all(valid_3d == count_gt) and np.all(valid_2d == count_gt)
    pts = np.asarray(pts.cpu() if hasattr(pts, "cpu") else pts)
    big = batch["pixel_gt_count"].numpy()[0, :, 0, 0] > 200
    assert np.abs(pts[0][big] - kp2[0][big][..., ::-1]).max() < 0.5          # RANSAC keypoints are (x,y)


def test_pvnet_separated_fields_drivers_and_evaluation_chain(device, tmp_path):
    """modelname pvnet (separated vector fields, train_casapose.py:221,313-320 / test_casapose.py:143): both drivers run end to end with
    estimate_confidence = estimate_coords = 0, and the evaluation chain for that output layout -- per-pixel selection of the arg-max object's
    slice (pose_evaluation.py:38-45), RANSAC voting, PnP, ADD -- recovers every visible object from ground-truth fields."""
    import test_casapose
    import train_casapose
    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset
    from casapose_amd.pose_estimation.pose_evaluation import estimate_and_evaluate_poses

    out = str(tmp_path / "run")
    names = "obj_000001,obj_000005,obj_000006"
    common = ["-c", CFG, "--outf", out, "--manualseed", "5", "--workers", "0", "--modelname", "pvnet", "--object", names, "--estimate_confidence", "0",
              "--estimate_coords", "0", "--confidence_regularization", "0", "--train_vectors_with_ground_truth", "0"]
    train_casapose.main(common + ["--data", "synthetic:4", "--datatest", "synthetic:2", "--epochs", "1", "--batchsize", "2", "--imagesize", "128", "160",
                                  "--saveinterval", "1", "--loginterval", "1", "--validationinterval", "1"])
    rows = list(csv.reader(open(out + "/loss_train.csv")))
    vals = np.array([[float(v) for v in r[2:7]] for r in rows[1:]])
    assert len(rows) == 3 and np.all(np.isfinite(vals)) and np.all(vals[:, 4] == 0)          # no keypoint loss with separated fields
    res = test_casapose.main(common + ["--datatest", "synthetic:2", "--load_h5_weights", "1", "--imagesize_test", "128", "160"])
    assert res["valid_3d"].shape == (3,) and np.all(np.isfinite(res["loss"]))
    # ground-truth separated fields -> 100 % recall
    oc, h, w, kp = 3, 448, 448, 9
    ds = SyntheticSceneDataset(oc, (h, w), length=1, seed=8)
    batch = ds.batch(0, 1)
    lab = batch["filtered_seg"][..., 0].numpy()
    kp2 = batch["target_vert"][:, :, 0].numpy()
    yy, xx = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    dirs = np.random.default_rng(0).standard_normal((1, h, w, oc, kp, 2)).astype(np.float32)  # foreign slices hold noise: they must not be read
    for o in range(oc):
        m = lab[0] == o + 1
        d = kp2[0, o][None, None] - np.stack([yy, xx], -1)[:, :, None, :]
        d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
        dirs[0, :, :, o][m] = d[m]
    seg = (10.0 * batch["target_seg"]).to(device)
    stats, poses, _ = estimate_and_evaluate_poses(seg, batch["target_seg"], torch.from_numpy(dirs.reshape(1, h, w, oc * kp * 2)).to(device), batch["poses_gt"],
                                                  batch["keypoints3d"], batch["cam_mat"], batch["diameters"], batch["offsets"],
                                                  evaluation_points=ds.mesh_vertex_array, object_points_3d_count=ds.mesh_vertex_count, min_num=200)
    valid_2d, valid_3d, count_gt = stats[0], stats[1], stats[2]
    assert count_gt.sum() >= 2 and np.all(valid_3d == count_gt) and np.all(valid_2d == count_gt)
