#!/usr/bin/env python3
"""Training driver with the reference's command line, config file, output layout and log formats
(fraunhoferhhi/casapose train_casapose.py:150-960), running on the MI355X engine.

    python train_casapose.py -c config/config_8.ini --data synthetic:64 --datatest synthetic:16 --epochs 2 --batchsize 4
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_casapose.py -c config/config_8.ini ...

Differences that are forced by this environment and documented in DESIGN.md:
  * multi-GPU = one process per GPU over torch.distributed/RCCL (SyncBN statistics + SUM gradient all-reduce, the
    semantics of MirroredStrategy) instead of one process driving all GPUs; `--batchsize` stays the GLOBAL batch;
  * `--data <folder>` reads an NDDS / converted-BOP tree like the reference (casapose_amd/data_handler/vectorfield_dataset.py);
    `--data synthetic[:N]` selects the built-in ray-cast scene generator (synthetic_scene.py) -- there is no dataset here;
  * checkpoints (`training_checkpoints/ckpt-<n>`, tf.train.Checkpoint in the reference) are .npz files holding the network only,
    like the reference's Checkpoint(network=net); `frozen_model/result_w.h5` is real HDF5 in Keras' save_weights layout
    (casapose_amd/utils/h5_weights.py).
"""
import datetime
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from casapose_amd import parallel  # noqa: E402
from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset  # noqa: E402
from casapose_amd.pose_estimation.pose_evaluation import estimate_and_evaluate_poses, evaluate_pose_estimates  # noqa: E402
from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted  # noqa: E402
from casapose_amd.pose_models.tfkeras import Classifiers  # noqa: E402
from casapose_amd.training import Adam, copy_weights_add_confidence_maps, copy_weights_from_backup_network, train_step  # noqa: E402
from casapose_amd.utils.config_parser import parse_config  # noqa: E402
from casapose_amd.utils.io_utils import latest_checkpoint  # noqa: E402
from casapose_amd.utils.learning_rate_schedules import ExponentialDecayLateStart, LossWeightHandler, PiecewiseConstantDecay  # noqa: E402


def create_dir(path):
    os.makedirs(path, exist_ok=True)


def open_dataset(spec, opt, no_objects, image_size, random_crop, seed):
    if spec == "":
        return None
    if not spec.startswith("synthetic"):  # an NDDS / converted-BOP folder tree (the reference's format)
        from casapose_amd.data_handler.vectorfield_dataset import VectorfieldDataset

        objs = [x.strip() for x in opt.object.split(",")]
        if random_crop:  # training set (train_casapose.py:226-250)
            return VectorfieldDataset(root=spec, path_meshes=opt.datameshes, path_filter_root=opt.data_path_filter, color_input=opt.color_dataset,
                                      no_points=opt.no_points, objectsofinterest=objs, noise=opt.noise, contrast=opt.contrast, brightness=opt.brightness,
                                      random_translation=(opt.translation, opt.translation), random_rotation=opt.rotation, random_crop=True,
                                      use_train_split=(opt.data == opt.datatest), train_validation_split=opt.train_validation_split,
                                      wxyz_quaterion_input=opt.data_wxyz_quaterion, seed=seed)
        return VectorfieldDataset(root=spec, path_meshes=opt.datameshes, path_filter_root=opt.datatest_path_filter, color_input=opt.color_dataset,
                                  no_points=opt.no_points, objectsofinterest=objs, noise=0.00001, contrast=0.00001, brightness=0.00001,
                                  random_translation=(0, 0), random_rotation=0, random_crop=False, use_validation_split=(opt.data == opt.datatest),
                                  train_validation_split=opt.train_validation_split, wxyz_quaterion_input=opt.datatest_wxyz_quaterion, seed=seed)
    n = int(spec.split(":")[1]) if ":" in spec else 64
    return SyntheticSceneDataset(no_objects, image_size, opt.no_points, length=n, seed=seed, random_crop=random_crop)


def pose_statistics(net, batch, opt, no_objects, device):
    """The pose_validation branch of train_step (train_casapose.py:651-676): voted keypoints -> poses -> per-object counts."""
    from casapose_amd.training import poses_from_coords

    img = batch["img"].to(device)
    seg = batch["target_seg"].to(device)
    inputs = [img, seg] if opt.train_vectors_with_ground_truth else [img]
    out = net(inputs, training=False)
    K, kp = no_objects + 1, opt.no_points
    o_seg, o_dirs, conf = torch.split(out, [K, 2 * kp, out.shape[3] - K - 2 * kp], dim=3)
    if opt.estimate_coords:
        coords = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=K, num_points=kp, filter_estimates=False)(
            [seg if opt.train_vectors_with_ground_truth else o_seg, o_dirs, conf])
        est = torch.argmax(o_seg, dim=3)
        avail = torch.stack([((est == o).sum(dim=(1, 2)) > 50) & ((seg[..., o] != 0).sum(dim=(1, 2)) > 50) for o in range(1, K)], dim=1)
        poses, pts = poses_from_coords(coords, avail, batch)
        stats, _, _ = evaluate_pose_estimates(pts, poses, batch["poses_gt"], seg, batch["keypoints3d"], batch["cam_mat"], batch["diameters"], min_num=200)
    else:
        stats, _, _ = estimate_and_evaluate_poses(o_seg, seg, o_dirs, batch["poses_gt"], batch["keypoints3d"], batch["cam_mat"], batch["diameters"],
                                                  batch["offsets"], min_num=200)
    return stats


def main(argv=None):
    print("start:", datetime.datetime.now().time())
    opt = parse_config(argv)
    rank, local, world = parallel.init_from_env("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("train_casapose.py needs a ROCm GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local if world > 1 else max(opt.gpuids[0], 0))
    device = torch.device("cuda", torch.cuda.current_device())
    checkpoint_path = opt.outf + "/" + opt.net
    frozen_path = opt.outf + "/frozen_model"
    if rank == 0:
        for p in (opt.outf, checkpoint_path, frozen_path):
            create_dir(p)
        with open(opt.outf + "/header.txt", "w") as f:
            f.write(str(opt))
    np.random.seed(opt.manualseed)
    torch.manual_seed(opt.manualseed)
    objectsofinterest = [x.strip() for x in opt.object.split(",")]
    no_objects = len(objectsofinterest)
    if opt.batchsize % world:
        raise SystemExit("--batchsize %d (global) must be divisible by the %d replicas" % (opt.batchsize, world))
    local_bs = opt.batchsize // world
    train_ds = open_dataset(opt.data, opt, no_objects, opt.imagesize, True, opt.manualseed)
    test_ds = open_dataset(opt.datatest, opt, no_objects, opt.imagesize, False, opt.manualseed + 1)
    # per-replica sharding AT THE SOURCE: each rank renders / reads only its slice of the global batch
    gen = lambda ds: ds.generate_dataset(opt.batchsize, opt.epochs, opt.prefetch, opt.imagesize, opt.crop_factor, opt.workers, no_objects,  # noqa: E731
                                         shard=(rank, world))
    trainingdata, train_batches = gen(train_ds) if train_ds else (None, 0)
    testingdata, test_batches = gen(test_ds) if test_ds else (None, 0)
    print("training data: {} batches".format(train_batches))
    print("testing data: {} batches".format(test_batches))

    height, width = opt.imagesize
    input_segmentation_shape = (height, width, 1 + no_objects) if opt.train_vectors_with_ground_truth else None
    separated_vectorfields = opt.modelname == "pvnet"      # train_casapose.py:221,313-320: one 2*points slice per object, no confidence / keypoint loss
    if separated_vectorfields and (opt.estimate_confidence or opt.estimate_coords) and no_objects > 1:
        raise SystemExit("modelname pvnet (separated vector fields) is not compatible with estimate_confidence / estimate_coords")
    ver_dim = opt.no_points * 2 * (no_objects if separated_vectorfields else 1) + (opt.no_points if opt.estimate_confidence else 0)
    net = Classifiers.get(opt.modelname)(ver_dim=ver_dim, seg_dim=1 + no_objects, input_shape=(height, width, 3),
                                         input_segmentation_shape=input_segmentation_shape, weights="imagenet" if opt.pretrained else None,
                                         base_model=opt.backbonename, device=device, seed=opt.manualseed)
    if opt.lr_epochs_steps is not None:
        boundaries = ((np.array(opt.lr_epochs_steps) * train_batches) - 1).tolist()
        values = (np.power(opt.lr_decay, np.arange(len(boundaries) + 1)) * opt.lr).tolist()
        lr_schedule = PiecewiseConstantDecay(boundaries, values)
    else:
        lr_schedule = ExponentialDecayLateStart(opt.lr, decay_steps=train_batches * opt.lr_epochs, decay_steps_start=train_batches * opt.lr_epochs_start,
                                                decay_rate=opt.lr_decay, staircase=True)
    optimizer = Adam(learning_rate=lr_schedule)
    net_backup = None
    ctor = Classifiers.get(opt.modelname)
    if opt.copy_weights_add_confidence_maps and opt.estimate_confidence:  # train_casapose.py:352-361
        net_backup = ctor(ver_dim=ver_dim - opt.no_points, seg_dim=1 + no_objects, input_shape=(height, width, 3),
                          input_segmentation_shape=input_segmentation_shape, weights=None, base_model=opt.backbonename, device=device)
    elif opt.copy_weights_from_backup_network:  # :362-370
        net_backup = ctor(ver_dim=ver_dim, seg_dim=1 + opt.objects_in_input_network, input_shape=(height, width, 3), input_segmentation_shape=None,
                          weights=None, base_model=opt.backbonename, device=device)
    if net_backup is not None:
        net_backup.load_weights(frozen_path + "/" + opt.load_h5_filename + ".h5", by_name=True, skip_mismatch=True)
        print("loaded backup network")
    save_count = [0]
    if opt.load_h5_weights:
        net.load_weights(frozen_path + "/" + opt.load_h5_filename + ".h5", by_name=True, skip_mismatch=True)
    elif opt.net != "":  # resume: checkpoint.restore(tf.train.latest_checkpoint(checkpoint_path)) (train_casapose.py:379-393)
        latest = latest_checkpoint(checkpoint_path)
        if latest is not None:
            net.load_weights(latest[0])
            save_count[0] = latest[1]  # tf.train.Checkpoint restores its save_counter: numbering continues
            print("restored {}".format(latest[0]))
        else:
            print("no checkpoint under {}: starting from the initial weights".format(checkpoint_path))
    if opt.copy_weights_add_confidence_maps and opt.estimate_confidence:
        copy_weights_add_confidence_maps(net, net_backup, ver_dim - opt.no_points)
    elif opt.copy_weights_from_backup_network:
        copy_weights_from_backup_network(net, net_backup, opt.objects_to_copy)
    del net_backup
    net.summary(print_fn=print if rank == 0 else (lambda *_: None))
    loss_factors = LossWeightHandler(mask_loss_weight=opt.mask_loss_weight, vertex_loss_weight=opt.vertex_loss_weight,
                                     proxy_loss_weight=opt.proxy_loss_weight, kp_loss_weight=opt.keypoint_loss_weight,
                                     filter_vertex_with_segmentation=opt.filter_vertex_with_segmentation,
                                     filter_high_proxy_errors=opt.filter_high_proxy_errors)
    if rank == 0:
        title = "epoch,batchid,loss,mask_loss,vertex_loss,proxy_loss,keypoint_loss,mask_loss_weight,vertex_loss_weight,proxy_loss_weight, kp_loss_weight\n"
        for name in ("/loss_train.csv", "/loss_test.csv"):
            with open(opt.outf + name, "w") as f:
                f.write(title)
        with open(opt.outf + "/train_summary.csv", "w") as f:
            f.write("epoch,learning_rate,loss,mask_loss,vertex_loss,proxy_loss,keypoint_loss\n")
        with open(opt.outf + "/test_summary.csv", "w") as f:
            s = "epoch,learning_rate,loss,mask_loss,vertex_loss,proxy_loss,keypoint_loss"
            s += "".join(",2d_{}".format(o) for o in objectsofinterest) + "".join(",3d_{}".format(o) for o in objectsofinterest)
            f.write(s + "\n")
    group = torch.distributed.group.WORLD if world > 1 else None

    def runnetwork(iterator, batches_per_epoch, epoch, train=True, pose_validation=False):
        lr = optimizer.lr
        epoch_loss = np.zeros(5)
        pose_acc = np.zeros((6, no_objects))
        start = time.time()
        for batch_idx in range(batches_per_epoch):
            batch = next(iterator)
            loss = train_step(net, batch, loss_factors, optimizer, opt, group, world, train=train)
            st = pose_statistics(net, batch, opt, no_objects, device) if (pose_validation and not train) else None
            # ONE packed collective per step for everything that is logged: the 5 loss scalars (MEAN over replicas, :690-694) and the
            # 6 per-object pose-statistic vectors (SUM over replicas, :732-737)
            loss, st = parallel.reduce_step_log(loss, st, world, device)
            if st is not None:
                pose_acc += st
            epoch_loss += np.array(loss[:5])
            if rank == 0:
                with open(opt.outf + ("/loss_train.csv" if train else "/loss_test.csv"), "a") as f:
                    f.write("{}, {},{:.15f},{:.7f},{:.7f},{:.7f},{:.7f},{:.4f},{:.4f},{:.4f},{:.4f}\n".format(
                        epoch, batch_idx + 1, loss[0], loss[1], loss[2], loss[3], loss[4], loss_factors.mask_loss_weight,
                        loss_factors.vertex_loss_weight, loss_factors.proxy_loss_weight, loss_factors.kp_loss_weight))
                if (batch_idx + 1) % opt.loginterval == 0:
                    print("{}  {} Epoch: {}, Batch idx: {}, Loss: {:.15f}, Epoch Loss: {:.15f}\n".format(
                        datetime.datetime.now().time(), "Train" if train else "Test", epoch, batch_idx + 1, loss[0], epoch_loss[0] / (batch_idx + 1)))
                    print("Time {}".format(time.time() - start))
            start = time.time()
        epoch_loss /= max(batches_per_epoch, 1)
        if rank != 0:
            return
        print("==========================")
        if train:
            print("== TRAINING == Finished epoch {} (lr={:.7f}) with total loss: {:.7f} --- mask: {:.7f}, vector: {:.7f}, proxy: {:.7f}, keypoint: {:.7f} ==".format(
                epoch, lr, *epoch_loss))
        else:
            print("== VALIDATION == Finished epoch {} with total loss: {:.7f} --- mask: {:.7f}, vector: {:.7f}, proxy: {:.7f}, keypoint: {:.7f} ==".format(
                epoch, *epoch_loss))
        err_2d = err_3d = None
        if pose_validation:
            gt = pose_acc[2]
            err_2d = np.divide(pose_acc[0], gt, out=np.zeros_like(gt), where=gt != 0)
            err_3d = np.divide(pose_acc[1], gt, out=np.zeros_like(gt), where=gt != 0)
            print("2D Valid: {}".format(err_2d))
            print("2D Valid (mean): {}".format(err_2d.mean()))
            print("3D Valid: {}".format(err_3d))
            print("3D Valid (mean): {}".format(err_3d.mean()))
            print("Err 2D: {}".format(np.divide(pose_acc[4], gt, out=np.zeros_like(gt), where=gt != 0)))
        loss_factors.print()
        print("==========================")
        with open(opt.outf + ("/train_summary.csv" if train else "/test_summary.csv"), "a") as f:
            s = "{},{},{:.7f},{:.7f},{:.7f},{:.7f},{:.7f}".format(epoch, lr, *epoch_loss)
            if pose_validation:
                s += "".join(",{:.4f}".format(v) for v in err_2d) + "".join(",{:.4f}".format(v) for v in err_3d)
            f.write(s + "\n")
        if epoch % opt.saveinterval == 0 and train:
            save_count[0] += 1
            net.save_weights(checkpoint_path + "/ckpt-%d.npz" % save_count[0])
            print("\nSave results weights as h5...\n")
            net.save_weights(frozen_path + "/result_w.h5")

    print("Batches per epoch: {} Epochs: {} : ".format(train_batches, opt.epochs))
    print("Test Batches per epoch: {} Epochs: {} : ".format(test_batches, opt.epochs))
    for epoch in range(1, opt.epochs + 1):
        if trainingdata is not None:
            runnetwork(trainingdata, int(train_batches), epoch, train=True)
        if testingdata is not None:
            runnetwork(testingdata, int(test_batches), epoch, train=False, pose_validation=(epoch % opt.validationinterval == 0))
    if rank == 0:
        save_count[0] += 1
        net.save_weights(checkpoint_path + "/ckpt-%d.npz" % save_count[0])
        net.save_weights(frozen_path + "/result_w.h5")
    print("end:", datetime.datetime.now().time())
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
