"""Host side of one training step: the counterpart of train_casapose.py:494-611 (`train_step` inside
`runnetwork`) and :334-347 (optimizer construction) for the MI355X engine.

    net = Classifiers.get("casapose_c_gcu5")(...)
    opt = Adam(learning_rate=PiecewiseConstantDecay(boundaries, values))
    losses = train_step(net, batch, loss_factors, opt, cfg)     # [loss, mask, vertex, proxy, kp] like compute_loss

The arithmetic runs in libcasapose_hip.so through casapose_amd.train_engine.TrainPlan; this module only maps the
reference's batch tuple and option names onto it.  Data parallelism: pass `group`/`world_size`
(torch.distributed over RCCL) -- SyncBN statistics and the flat gradient are SUM-all-reduced
(MirroredStrategy semantics, train_casapose.py:195,641-643).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import train_engine as TE


class Adam:
    """tf.keras.optimizers.Adam surface used by the reference (learning_rate float or schedule, beta_1, beta_2,
    epsilon=1e-7, `.iterations`, `.lr` / `._decayed_lr`)."""

    def __init__(self, learning_rate: Union[float, Callable[[int], float]] = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, name: str = "Adam"):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon, self.name = learning_rate, beta_1, beta_2, epsilon, name
        self.iterations = 0

    def current_lr(self) -> float:
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    lr = property(current_lr)

    def apply(self, store: TE.ParamStore, stream: int):
        store.step_count = self.iterations
        store.adam_step(self.current_lr(), stream, self.beta_1, self.beta_2, self.epsilon)
        self.iterations += 1


def labels_from_onehot(seg: torch.Tensor) -> torch.Tensor:
    """[B,H,W,K] one-hot / score map (or [B,H,W(,1)] integer map) -> uint8 label map (tf.argmax, first maximum)."""
    if seg.dim() == 4 and seg.shape[-1] > 1:
        return torch.argmax(seg, dim=-1).to(torch.uint8).contiguous()
    if seg.dim() == 4:
        seg = seg[..., 0]
    return seg.to(torch.uint8).contiguous()


def _host(x, dtype=np.float64):
    return np.asarray(x.cpu() if torch.is_tensor(x) else x, dtype)


def _kp_targets(batch, B, oc, dev):
    cam = _host(batch["cam_mat"])
    cam = cam[0] if cam.ndim == 3 else cam  # the reference uses camera_data[0] for the whole batch (loss_functions.py:322)
    p3d = _host(batch["keypoints3d"]).reshape(B, oc, -1, 3)
    poses = _host(batch["poses_gt"]).reshape(B, oc, 3, 4)
    gt_xy = torch.from_numpy(TE.project_keypoints(p3d, cam, poses)).to(dev).contiguous()
    aff = torch.from_numpy(TE.crop_to_image_affine(_host(batch["offsets"]))).to(dev).contiguous()
    return gt_xy, aff, cam, p3d


def bpnp_reprojection_loss_host(coords_yx, gt_xy, affine, avail, points_3d, cam, max_pixel_error: float = 25.0, weight: float = 1.0, rng=None):
    """keypoint_reprojection_loss with use_bpnp_reprojection_loss=True (loss_functions.py:264-323), evaluated on the host in
    fp64 like the reference's BPNP_fast (host PnP inside tf.numpy_function + implicit-function gradient):
        p' = T(p)  (crop -> image), pose = PnP(p'), r = project(points_3d, pose)
        loss = sum_n avail * mean_j cap(smoothL1((|r - p'| + |gt - r|) / 2)) / sum(avail)
    Returns (loss, g_yx = weight * d loss / d coords_yx [B,oc,kp,2], poses [B,oc,1,3,4])."""
    from .pose_estimation import pnp as _pnp

    c = _host(coords_yx)
    B, oc, kp, _ = c.shape
    gt = _host(gt_xy).reshape(B, oc, kp, 2)
    A = _host(affine).reshape(B, 2, 3)
    av = _host(avail).reshape(B, oc)
    X = _host(points_3d).reshape(B, oc, kp, 3)
    K = _host(cam)
    na = av.sum()
    g = np.zeros((B, oc, kp, 2))
    poses = np.zeros((B, oc, 1, 3, 4), np.float32)
    total = 0.0
    for n in range(B):
        L = A[n, :, :2]                                   # d(X,Y)/d(x,y)
        for o in range(oc):
            if av[n, o] == 0:
                continue
            xy = c[n, o, :, ::-1] @ L.T + A[n, :, 2]      # image pixels (x,y)
            p6 = _pnp.pnp_rvec_t(X[n, o], xy, K, rng=rng).astype(np.float64)
            if not np.all(np.isfinite(p6)):
                raise FloatingPointError("PnP returned a non-finite pose (image %d, object %d)" % (n, o))
            rv, t = _pnp.refine_lm(X[n, o], xy, K, p6[:3], p6[3:], iters=30, eps=1e-14)   # tighten the optimum the IFT differentiates
            p6 = np.concatenate([rv, t])
            R = _pnp.rodrigues(rv)
            P = np.concatenate([R, t.reshape(3, 1)], axis=1)
            poses[n, o, 0] = -P if t[2] < 0 else P
            res, J = _pnp._residual_and_jacobian(X[n, o], xy, K, rv, t)   # res = r - p' (interleaved u,v), J = d r / d pose6
            r = res.reshape(kp, 2) + xy
            d1 = r - xy
            d2 = gt[n, o] - r
            n1 = np.sqrt((d1 ** 2).sum(1))
            n2 = np.sqrt((d2 ** 2).sum(1))
            e = 0.5 * (n1 + n2)
            l = np.where(e < 1.0, 0.5 * e * e, e - 0.5)
            slope = np.where(e < 1.0, e, 1.0)
            capped = l > max_pixel_error
            l = np.where(capped, max_pixel_error + (l - max_pixel_error) * 0.01, l)
            slope = np.where(capped, slope * 0.01, slope)
            total += l.mean()
            if na <= 0:
                continue
            cf = slope / (kp * na) * 0.5                                   # d loss / d n1 = d loss / d n2
            u1 = np.where(n1[:, None] > 0, d1 / np.where(n1[:, None] > 0, n1[:, None], 1.0), 0.0)
            u2 = np.where(n2[:, None] > 0, d2 / np.where(n2[:, None] > 0, n2[:, None], 1.0), 0.0)
            g_r = cf[:, None] * (u1 - u2)                                   # d loss / d r
            g_p_direct = -cf[:, None] * u1                                  # d loss / d p' (explicit dependence of |r - p'|)
            g_pose = J.T @ g_r.reshape(-1)
            g_p = g_p_direct + _pnp.bpnp_backward(g_pose, xy, X[n, o], K, p6)
            g[n, o] = (g_p @ L)[:, ::-1]                                   # back through the affine, then (x,y) -> (y,x)
    loss = total / na if na > 0 else 0.0
    return float(loss), (weight * g).astype(np.float32), poses


def train_step(net, batch: Dict[str, torch.Tensor], loss_factors, optimizer: Optional[Adam], opt, group=None, world_size: int = 1,
               train: bool = True, coords: Optional[torch.Tensor] = None, min_num: int = 50, min_num_gt: Optional[int] = None,
               filter_with_gt: bool = True):
    """One step on a batch dict with the reference's tuple fields (train_casapose.py:496-507):
       img [B,H,W,3]; target_seg [B,H,W,K] one-hot (or a uint8 label map); keypoints3d [B,oc,1,kp,3]; target_vert
       [B,oc,1,kp,2] 2-D keypoints (y,x) in crop pixels; cam_mat [B,3,3] or [3,3]; offsets [B,10]; filtered_seg
       (optional label map); poses_gt [B,oc,1,3,4].
    `opt` carries the config flags (train_vectors_with_ground_truth, estimate_coords, max_keypoint_pixel_error,
    confidence_regularization, use_bpnp_reprojection_loss).  train=True: forward with batch statistics, backward, Adam.
    train=False: the network runs in inference mode (moving statistics, net(net_input, training=False),
    train_casapose.py:596) and only the loss values are computed; filtered_seg is ignored like in the reference's
    evaluation branch (:637-648).  Returns python floats [loss, mask_loss, vertex_loss, proxy_loss, kp_loss]
    (compute_loss, train_casapose.py:137-145)."""
    plan, dev = net.training_plan(batch["img"].shape[0], batch["img"].shape[1], batch["img"].shape[2], group, world_size)
    img = batch["img"].to(device=dev, dtype=torch.float32).contiguous()
    seg_in = batch["target_seg"].to(dev)
    labels = labels_from_onehot(seg_in)
    fg = labels_from_onehot(batch["filtered_seg"].to(dev)) if (train and batch.get("filtered_seg") is not None) else labels
    kpts = batch["target_vert"].to(device=dev, dtype=torch.float32)
    B, oc = kpts.shape[0], kpts.shape[1]
    kpts = kpts.reshape(B, oc, -1, 2).contiguous()
    with_gt = bool(getattr(opt, "train_vectors_with_ground_truth", False))
    if train:
        plan.update_moving = True
        plan.forward(img, labels if with_gt else None)
    else:
        if with_gt and seg_in.dim() != 4:
            seg_in = torch.nn.functional.one_hot(labels.long(), net.seg_dim)
        out = net([img, seg_in.to(torch.float32)] if with_gt else [img], training=False)
        plan.out_view.copy_(out)
    wts = (float(loss_factors.mask_loss_weight), float(loss_factors.vertex_loss_weight), float(loss_factors.proxy_loss_weight))
    sums = plan.loss_and_grad(labels, fg, kpts, *wts, filter_with_segmentation=bool(loss_factors.filter_vertex_with_segmentation),
                              filter_high_proxy_errors=bool(getattr(loss_factors, "filter_high_proxy_errors", False)) and train and batch.get("pixel_gt_count") is not None)
    kp_w = float(getattr(loss_factors, "kp_loss_weight", 0.0))
    kp_loss = None
    if getattr(opt, "estimate_coords", False):
        gt_xy, aff, cam, p3d = _kp_targets(batch, B, oc, dev)
        host_loss = None
        if getattr(opt, "use_bpnp_reprojection_loss", False):
            def host_loss(c, av):  # noqa: E306
                return bpnp_reprojection_loss_host(c, gt_xy, aff, av, p3d, cam, float(getattr(opt, "max_keypoint_pixel_error", 25.0)), kp_w)[:2]
        kp_loss = plan.kp_loss_and_grad(labels, gt_xy, aff, kp_w, max_pixel_error=float(getattr(opt, "max_keypoint_pixel_error", 25.0)), min_num=min_num,
                                        confidence_regularization=bool(getattr(opt, "confidence_regularization", False)) and train,
                                        vote_with_gt=with_gt, min_num_gt=min_num_gt, filter_with_gt=filter_with_gt, coords=coords, backward=train,
                                        host_loss=host_loss)
    if train:
        stream = torch.cuda.current_stream(dev).cuda_stream
        plan.backward()
        plan.all_reduce_grads()
        # layer.trainable = False (Keras surface): the variables of that layer are not in `trainable_variables`, so
        # apply_gradients never touches them or their moments -- the flat Adam launch runs over everything, frozen slices are restored
        st = plan.store
        frozen = [key for layer in net.layers if not layer.trainable for key in layer._keys if key in st.offsets]
        saved = [(key, st.view(key).clone(), st.view(key, st.m).clone(), st.view(key, st.v).clone()) for key in frozen]
        optimizer.apply(plan.store, stream)
        for key, p_, m_, v_ in saved:
            st.view(key).copy_(p_)
            st.view(key, st.m).copy_(m_)
            st.view(key, st.v).copy_(v_)
        plan.refresh_weights(stream)
        net.mark_trained()
    s = sums.cpu().numpy()
    kpv = float(kp_loss.item()) if kp_loss is not None else 0.0
    total = wts[0] * s[0] + wts[1] * s[1] + wts[2] * s[2] + kp_w * kpv
    return [float(total), float(s[0]), float(s[1]), float(s[2]), kpv]


def poses_from_coords(coords_yx, objects_available, batch, rng=None) -> Tuple[np.ndarray, np.ndarray]:
    """The pose branch of keypoint_reprojection_loss (loss_functions.py:229,264-315): voted keypoints (y,x) in crop
    pixels -> (x,y) -> original image pixels (transform_points_back) -> host PnP -> pose negated if t_z < 0 -> zeroed for
    unavailable objects.  Returns (poses [B,oc,1,3,4], image-space points [B,oc,kp,2]) as float32 arrays."""
    from .pose_estimation import pnp as _pnp
    from .pose_estimation.pose_evaluation import transform_points_back

    c = _host(coords_yx)
    B, oc, kp, _ = c.shape
    avail = _host(objects_available).reshape(B, oc)
    offs = _host(batch["offsets"])
    cam = _host(batch["cam_mat"])
    cam = cam[0] if cam.ndim == 3 else cam
    p3d = _host(batch["keypoints3d"]).reshape(B, oc, -1, 3)
    poses = np.zeros((B, oc, 1, 3, 4), np.float32)
    pts = np.zeros((B, oc, kp, 2), np.float32)
    for n in range(B):
        for o in range(oc):
            xy = transform_points_back(c[n, o, :, ::-1], offs[n])
            if avail[n, o] == 0:
                continue
            pts[n, o] = xy
            p6 = _pnp.pnp_rvec_t(p3d[n, o], xy, cam, rng=rng)
            if not np.all(np.isfinite(p6)):
                raise FloatingPointError("PnP returned a non-finite pose (image %d, object %d)" % (n, o))  # tf.Assert, loss_functions.py:299-304
            R, t = _pnp.rodrigues(p6[:3]), p6[3:].astype(np.float64)
            P = np.concatenate([R, t.reshape(3, 1)], axis=1)
            poses[n, o, 0] = -P if t[2] < 0 else P
    return poses, pts


def test_step(net, batch, opt, loss_factors, evaluation_points=None, object_points_3d_count=None):
    """One evaluation step of test_casapose.py (:268-384): inference forward, (component-filtered) LS keypoint voting or
    RANSAC voting, host PnP, losses and the eight per-object pose statistics.
    Returns (losses [5 floats], pose_stats [8 arrays of oc], poses [B,oc,3,4], points, seconds)."""
    import time

    from .pose_estimation.pose_evaluation import estimate_and_evaluate_poses, evaluate_pose_estimates
    from .pose_estimation.voting_layers_2d import CoordLSVotingWeighted

    dev = net.device
    img = batch["img"].to(device=dev, dtype=torch.float32).contiguous()
    seg = batch["target_seg"].to(device=dev, dtype=torch.float32)
    K, kp = net.seg_dim, int(getattr(opt, "no_points", 9))
    with_gt = bool(getattr(opt, "train_vectors_with_ground_truth", False))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = net([img, seg] if with_gt else [img], training=False)
    if K > 2 and out.shape[3] - K == 2 * kp * (K - 1):    # `pvnet`, separated vector fields: no confidences, RANSAC voting on per-object slices
        o_seg, o_dirs, conf = out[..., :K], out[..., K:], None
    else:
        o_seg, o_dirs, conf = torch.split(out, [K, 2 * kp, out.shape[3] - K - 2 * kp], dim=3)
    coords = None
    if getattr(opt, "estimate_coords", False):
        voter = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=K, num_points=kp,
                                      filter_estimates=bool(getattr(opt, "confidence_filter_estimates", True)),
                                      output_second_largest_component=bool(getattr(opt, "confidence_choose_second", False)))
        coords = voter([seg if with_gt else o_seg, o_dirs, conf])
    losses = train_step(net, batch, loss_factors, None, opt, train=False, coords=coords, min_num=int(getattr(opt, "min_object_size_test", 1)),
                        min_num_gt=1, filter_with_gt=bool(getattr(opt, "filter_test_with_gt", False)))
    if coords is not None:
        plan, _ = net.training_plan(img.shape[0], img.shape[1], img.shape[2])
        poses, pts = poses_from_coords(coords, plan.objects_available, batch)
        stats, poses, pts = evaluate_pose_estimates(pts, poses, batch["poses_gt"], seg, batch["keypoints3d"], batch["cam_mat"], batch["diameters"],
                                                    evaluation_points=evaluation_points, object_points_3d_count=object_points_3d_count, min_num=1)
        poses = np.asarray(poses).reshape(poses.shape[0], poses.shape[1], 3, 4)
    else:
        stats, poses, pts = estimate_and_evaluate_poses(o_seg, seg, o_dirs, batch["poses_gt"], batch["keypoints3d"], batch["cam_mat"], batch["diameters"],
                                                        batch["offsets"], evaluation_points=evaluation_points, object_points_3d_count=object_points_3d_count,
                                                        min_num=1)
    torch.cuda.synchronize(dev)
    return losses, stats, poses, pts, time.perf_counter() - t0


# ------------------------------------------------------------------------------------------------
# weight surgery of the training script (train_casapose.py:352-448)
# ------------------------------------------------------------------------------------------------
def copy_weights_add_confidence_maps(net, net_backup, ver_dim_backup: int, print_fn=print):
    """A network trained WITHOUT confidence maps initialises one with them: the first ver_dim_backup output channels of
    pv_final_conv_vertex are copied (copy_weights_vertex, :398-405)."""
    name = "pv_final_conv_vertex"
    print_fn("Copy weights for {}".format(name))
    block = net.get_layer(name).get_weights()
    backup = net_backup.get_layer(name).get_weights()
    block[0][0, 0, :, :ver_dim_backup] = backup[0][0, 0, :, :ver_dim_backup]
    net.get_layer(name).set_weights(block)
    return net


def copy_weights_from_backup_network(net, net_backup, objects_to_copy, print_fn=print):
    """Initialise a network for a new object set from one trained on another set: class-indexed weights (segmentation head columns,
    CLADE gamma / beta rows) are copied for the class pairs (in, out) of `objects_to_copy` (row 0 = background), everything
    class-independent comes from load_weights(by_name) (:407-442)."""
    table = np.asarray(objects_to_copy, dtype=np.int64).reshape(-1, 2)
    range_in, range_out = table[:, 0].tolist(), table[:, 1].tolist()
    name = "pv_final_conv_segmentation"
    print_fn("Copy weights for {}".format(name))
    block = net.get_layer(name).get_weights()
    backup = net_backup.get_layer(name).get_weights()
    block[0][0, 0, :, range_out] = backup[0][0, 0, :, range_in]
    net.get_layer(name).set_weights(block)
    for i in range(6, 11):
        name = "pv_block_%d_clade" % i
        print_fn("Copy weights for {}".format(name))
        block = net.get_layer(name).get_weights()
        backup = net_backup.get_layer(name).get_weights()
        for j in range(len(block)):
            if block[j].ndim == 2:  # gamma / beta tables [classes, channels]
                block[j][range_out] = backup[j][range_in]
        net.get_layer(name).set_weights(block)
    return net
