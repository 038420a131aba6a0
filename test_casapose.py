#!/usr/bin/env python3
"""Evaluation driver with the reference's command line, config file and report formats
(fraunhoferhhi/casapose test_casapose.py:60-600) on the MI355X engine: batch size 1 at `imagesize_test`, inference forward,
(component-filtered) LS keypoint voting or RANSAC voting, host PnP, ADD / ADD-S / 2-D projection recall per object.

    python test_casapose.py -c config/config_8.ini --datatest synthetic:32 --load_h5_weights 1 --load_h5_filename result_w

`--datatest <folder>` reads an NDDS / converted-BOP tree; `--datatest synthetic[:N]` selects the built-in scene generator
(no dataset on this machine).  Writes <evalf>/loss_test_eval.csv and <evalf>/test_summary_eval.csv with the reference's columns and, with
--write_poses, <evalf>/poses_out/bop_evaluation.csv plus the per-object pose text files (casapose_amd/utils/io_utils.py).
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset  # noqa: E402
from casapose_amd.pose_models.tfkeras import Classifiers  # noqa: E402
from casapose_amd.training import test_step  # noqa: E402
from casapose_amd.utils.config_parser import parse_config  # noqa: E402
from casapose_amd.utils.io_utils import latest_checkpoint, write_poses  # noqa: E402
from casapose_amd.utils.learning_rate_schedules import LossWeightHandler  # noqa: E402


def main(argv=None):
    opt = parse_config(argv)
    if not torch.cuda.is_available():
        raise SystemExit("test_casapose.py needs a ROCm GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(max(opt.gpuids[0], 0))
    device = torch.device("cuda", torch.cuda.current_device())
    checkpoint_path = opt.outf + "/" + opt.net
    frozen_path = opt.outf + "/frozen_model"
    os.makedirs(opt.evalf, exist_ok=True)
    objectsofinterest = [x.strip() for x in opt.object.split(",")]
    no_objects = len(objectsofinterest)
    height, width = opt.imagesize_test
    if opt.datatest.startswith("synthetic"):
        n = int(opt.datatest.split(":")[1]) if ":" in opt.datatest else 32
        ds = SyntheticSceneDataset(no_objects, (height, width), opt.no_points, length=n, seed=(opt.manualseed or 0) + 1, random_crop=False)
    else:  # an NDDS / converted-BOP folder tree, read like test_casapose.py:150-184
        from casapose_amd.data_handler.vectorfield_dataset import VectorfieldDataset

        ds = VectorfieldDataset(root=opt.datatest, path_meshes=opt.datameshes, path_filter_root=opt.datatest_path_filter, color_input=opt.color_dataset,
                                no_points=opt.no_points, objectsofinterest=objectsofinterest, noise=0.00001, contrast=0.00001, brightness=0.00001,
                                random_translation=(0, 0), random_rotation=0, random_crop=False, wxyz_quaterion_input=opt.datatest_wxyz_quaterion)
    testingdata, test_batches = ds.generate_dataset(1, 1, 0, opt.imagesize_test, 1.0, 1, no_objects, shuffle=False)
    mesh_vertex_array, mesh_vertex_count = ds.generate_object_vertex_array()
    print("testing data: {} batches".format(test_batches))
    input_segmentation_shape = (height, width, 1 + no_objects) if opt.train_vectors_with_ground_truth else None
    separated_vectorfields = opt.modelname == "pvnet"      # test_casapose.py:143,203-210: one 2*points slice per object, no confidences
    if separated_vectorfields and (opt.estimate_confidence or opt.estimate_coords) and no_objects > 1:
        raise SystemExit("modelname pvnet (separated vector fields) is not compatible with estimate_confidence / estimate_coords")
    ver_dim = opt.no_points * 2 * (no_objects if separated_vectorfields else 1) + (opt.no_points if opt.estimate_confidence else 0)
    net = Classifiers.get(opt.modelname)(ver_dim=ver_dim, seg_dim=1 + no_objects, input_shape=(height, width, 3),
                                         input_segmentation_shape=input_segmentation_shape, weights="imagenet" if opt.pretrained else None,
                                         base_model=opt.backbonename, device=device, seed=opt.manualseed)
    if opt.load_h5_weights:
        net_path = frozen_path + "/" + opt.load_h5_filename + ".h5"
        print(net_path)
        net.load_weights(net_path, by_name=True, skip_mismatch=True)
    elif opt.net != "":
        latest = latest_checkpoint(checkpoint_path)
        if latest is not None:
            net.load_weights(latest[0])
    for layer in net.layers:
        layer.trainable = False
    net.summary()
    with open(opt.evalf + "/loss_test_eval.csv", "w") as f:
        f.write("batchid,loss,mask_loss,vertex_loss,proxy_loss,kp_loss,mask_loss_weight,vertex_loss_weight,proxy_loss_weight,kp_loss_weight\n")
    with open(opt.evalf + "/test_summary_eval.csv", "w") as f:
        s = "loss,mask_loss,vertex_loss,proxy_loss,kp_loss,time"
        s += "".join(",2d_{}".format(o) for o in objectsofinterest) + ",2d_mean"
        s += "".join(",3d_{}".format(o) for o in objectsofinterest) + ",3d_mean\n"
        f.write(s)
    if os.path.exists(opt.evalf + "/poses_out/"):
        for fn in sorted(glob.glob(opt.evalf + "/poses_out/*/*.txt")):
            os.remove(fn)
    loss_factors = LossWeightHandler(opt.mask_loss_weight, opt.vertex_loss_weight, opt.proxy_loss_weight, opt.keypoint_loss_weight)

    test_loss = np.zeros(5)
    acc = {k: np.zeros(no_objects) for k in ("2d", "3d", "gt", "fp", "e2", "e3", "miss")}
    total_time = 0.0
    print("Test Batches: {} ".format(test_batches))
    for batch_idx in range(int(test_batches)):
        batch = next(testingdata)
        loss, st, poses, pts, seconds = test_step(net, batch, opt, loss_factors, evaluation_points=mesh_vertex_array, object_points_3d_count=mesh_vertex_count)
        valid_2d, valid_3d, count_gt, fp_mask, err_2d, err_3d, missing, fp_pose = [np.asarray(v, np.float64) for v in st]
        acc["2d"] += valid_2d
        acc["3d"] += valid_3d
        acc["gt"] += count_gt
        acc["fp"] += fp_pose
        acc["e2"] += err_2d
        acc["e3"] += err_3d
        acc["miss"] += missing
        test_loss += np.array(loss)
        total_time += seconds
        with open(opt.evalf + "/loss_test_eval.csv", "a") as f:
            f.write("{},{:.15f},{:.7f},{:.7f},{:.7f},{:.7f},{:.7f}\n".format(batch_idx + 1, loss[0], loss[1], loss[2], loss[3], loss[4], seconds))
        print("Batch idx: {}, Loss: {:.5f} --- mask: {:.5f}, vector: {:.5f}, proxy: {:.5}, kp: {:.5} -- Average Loss: {:.5f}\n".format(
            batch_idx, loss[0], loss[1], loss[2], loss[3], loss[4], test_loss[0] / (batch_idx + 1)))
        print("Test GT: {}".format(count_gt))
        print("Test 2D: {}".format(valid_2d))
        print("Test 3D: {}".format(valid_3d))
        print("Test Sum GT: {}".format(acc["gt"]))
        print("Test Sum 2D: {}".format(acc["2d"]))
        print("Test Sum 3D: {}".format(acc["3d"]))
        print("Misses: {}".format(acc["miss"]))
        print("False positive: {}".format(acc["fp"]))
        print("Err 2D: {}".format(err_2d))
        print("Err 3D: {}".format(err_3d))
        if opt.write_poses:
            image_id = batch.get("image_id", ["synthetic_%06d_%06d" % (0, batch_idx)])  # batch tuple entry 12 (vectorfield_dataset.py:309)
            write_poses(batch["poses_gt"][0], np.asarray(poses)[0], objectsofinterest, image_id, opt.evalf + "/poses_out/", seconds)
    test_loss /= max(test_batches, 1)
    gt = acc["gt"]
    div = lambda a, b: np.divide(a, b, out=np.zeros_like(a), where=b != 0)  # noqa: E731  (divide_no_nan)
    err_2d, err_3d = div(acc["2d"], gt), div(acc["3d"], gt)
    detection_count = np.where(gt == 0.0, 0.0, gt - acc["miss"] + acc["fp"])
    precision = div(acc["3d"], detection_count)
    print("==========================")
    print("== TEST == Finished test with total loss: {:.7f} --- mask: {:.7f}, vector: {:.7f}, proxy: {:.7f}, kp: {:.7} ==".format(*test_loss))
    print("2D Valid: {}".format(err_2d))
    print("2D Valid (mean): {}".format(err_2d.mean()))
    print("3D Valid: {}".format(err_3d))
    print("3D Valid (mean): {}".format(err_3d.mean()))
    print("3D Valid (precision): {}".format(precision))
    print("3D Valid (average precision): {}".format(precision.mean()))
    print("mean time per image: {:.4f} s".format(total_time / max(test_batches, 1)))
    print("==========================")
    with open(opt.evalf + "/test_summary_eval.csv", "a") as f:
        s = "{:.7f},{:.7f},{:.7f},{:.7f},{:.7f}".format(*test_loss)
        s += "".join(",{:.4f}".format(v) for v in err_2d) + ",{:.4f}".format(err_2d.mean())
        s += "".join(",{:.4f}".format(v) for v in err_3d) + ",{:.4f}".format(err_3d.mean())
        f.write(s + "\n")
    return {"loss": test_loss, "valid_2d": err_2d, "valid_3d": err_3d, "precision": precision}


if __name__ == "__main__":
    main()
