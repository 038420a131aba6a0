"""The reference's loss functions under their own names and argument lists (casapose/utils/loss_functions.py:14-344), evaluated by
libcasapose_hip.so (cp_smooth_l1_f32, cp_proxy_voting_f32, cp_kp_stats_f32, cp_kp_reproj_loss_f32; host PnP for the BPnP variant, like the
reference's BPNP_fast).  VALUES only: the training step differentiates the fused kernels of the same formulas (cp_pose_loss_f32,
cp_pose_loss_sep_f32, cp_ls_vote_bwd_f32), so these functions are for evaluation scripts and for code written against the reference's API.
Device tensors in, device tensors (or python floats where the reference returns scalars) out; no CPU fallback.

Tensor conventions are the reference's: vertex_pred / vertex_targets [B,H,W,C]; vertex_weights [B,H,W,1]; vertex_one_hot_weights [B,H,W,oc]
(target_seg[..., 1:]); keypoint_targets [B,oc,ic,kp,2] (y,x).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import check


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(t):
    if not t.is_cuda:
        raise _lib.CasaposeHipError("casapose.utils.loss_functions needs CUDA (ROCm) tensors; there is no CPU fallback")


def _f32(t, dev=None):
    return t.to(device=dev if dev is not None else t.device, dtype=torch.float32).contiguous()


def _wmode(ignore_weights: bool, invert_weights: bool) -> int:
    return 2 if ignore_weights else (1 if invert_weights else 0)


def smooth_l1_loss(vertex_pred, vertex_targets, vertex_weights, ignore_weights=False, invert_weights=False, normalize=True, reduce=True):
    """loss_functions.py:14-44: smoothL1(|w (pred - target)|); normalize: per image sum / (C sum(w) + 1e-3); reduce: mean."""
    _need_cuda(vertex_pred)
    pred, tgt, wts = _f32(vertex_pred), _f32(vertex_targets, vertex_pred.device), _f32(vertex_weights, vertex_pred.device)
    b, h, w, c = pred.shape
    sums = torch.empty(b, 2, dtype=torch.float64, device=pred.device)
    elem = torch.empty(b, h, w, c, dtype=torch.float32, device=pred.device) if not normalize else None
    check(_lib.load().cp_smooth_l1_f32(pred.data_ptr(), c, tgt.data_ptr(), tgt.shape[-1], wts.data_ptr(), wts.shape[-1] if wts.dim() == 4 else 1,
                                       _wmode(ignore_weights, invert_weights), c, b, h * w, sums.data_ptr(), elem.data_ptr() if elem is not None else None,
                                       _stream(pred)), "cp_smooth_l1_f32")
    if normalize:
        per_image = (sums[:, 0] / (c * sums[:, 1] + 1e-3)).to(torch.float32)
        return per_image.mean() if reduce else per_image
    return (sums[:, 0].sum() / (b * h * w * c)).to(torch.float32) if reduce else elem


def _object_labels(one_hot: torch.Tensor) -> torch.Tensor:
    """[B,H,W,oc] one-hot rows -> uint8 map: 0 where the row is empty, o+1 where channel o is set (format adaptation only)."""
    oh = one_hot.to(torch.float32)
    return torch.where(oh.sum(-1) > 0, torch.argmax(oh, dim=-1) + 1, torch.zeros((), dtype=torch.int64, device=oh.device)).to(torch.uint8).contiguous()


def _proxy(vertex_pred, keypoint_targets, vertex_one_hot_weights, vertex_weights, invert_weights, want_objects, want_dist, want_elem):
    _need_cuda(vertex_pred)
    pred = _f32(vertex_pred)
    dev = pred.device
    b, h, w, ver_dim = pred.shape
    oc = vertex_one_hot_weights.shape[-1]
    kt = _f32(keypoint_targets, dev)
    _, koc, ic, kp, _ = kt.shape
    if koc != oc or ver_dim != 2 * kp:
        raise ValueError("proxy voting: %d direction channels for %d keypoints, %d mask channels for %d keypoint sets" % (ver_dim, kp, oc, koc))
    labels = _object_labels(vertex_one_hot_weights.to(dev))
    wts = _f32(vertex_weights, dev)
    img = torch.empty(b, 2, dtype=torch.float64, device=dev)
    osum = torch.empty(b, oc, dtype=torch.float64, device=dev) if want_objects else None
    ocnt = torch.empty(b, oc, dtype=torch.int32, device=dev) if want_objects else None
    dist = torch.empty(b, h, w, kp, dtype=torch.float32, device=dev) if want_dist else None
    elem = torch.empty(b, h, w, kp, dtype=torch.float32, device=dev) if want_elem else None
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(_lib.load().cp_proxy_voting_f32(pred.data_ptr(), ver_dim, kp, labels.data_ptr(), wts.data_ptr(), wts.shape[-1] if wts.dim() == 4 else 1,
                                          1 if invert_weights else 0, kt.data_ptr(), oc, ic, b, h, w, img.data_ptr(), p(osum), p(ocnt), p(dist), p(elem),
                                          _stream(pred)), "cp_proxy_voting_f32")
    return img, osum, ocnt, dist, elem, (b, h, w, ver_dim, oc, kp)


def _select_object_slices(vertex_pred, vertex_one_hot_weights, vertex_weights, object_count, keypoint_count):
    """proxy_voting_dist's first branch (loss_functions.py:59-82): a separated field [B,H,W,oc*kp*2] is reduced to the slice of each pixel's
    own object (zero where vertex_weights > 0, i.e. on the background) -- an indexing step, done with torch indexing."""
    b, h, w, _ = vertex_pred.shape
    v = vertex_pred.reshape(b, h, w, object_count, keypoint_count * 2)
    idx = torch.argmax(vertex_one_hot_weights.to(torch.float32), dim=3)
    v = torch.gather(v, 3, idx[..., None, None].expand(b, h, w, 1, keypoint_count * 2))[:, :, :, 0]
    return torch.where(vertex_weights.to(v.device) > 0, torch.zeros((), dtype=v.dtype, device=v.device), v)


def proxy_voting_dist(vertex_pred, keypoint_targets, vertex_one_hot_weights, vertex_weights, invert_weights=False, min_object_pixel=20):
    """loss_functions.py:47-129 -> (dist [B,H,W,kp], per-object loss [B,oc]): dist = |w * perpendicular distance keypoint <-> pixel ray|;
    per object: sum of smoothL1(dist) over its pixels / (kp * pixels + 1e-3), zero for objects with fewer than min_object_pixel pixels."""
    oc = vertex_one_hot_weights.shape[-1]
    kp = keypoint_targets.shape[3]
    if oc > 1 and vertex_pred.shape[-1] == oc * kp * 2:
        vertex_pred = _select_object_slices(vertex_pred, vertex_one_hot_weights, vertex_weights, oc, kp)
    img, osum, ocnt, dist, _, (b, h, w, ver_dim, oc, kp) = _proxy(vertex_pred, keypoint_targets, vertex_one_hot_weights, vertex_weights, invert_weights, True, True, False)
    cnt = ocnt.to(torch.float64)
    valid = (cnt >= min_object_pixel).to(torch.float64)
    in_loss = valid * osum / ((ver_dim / 2) * cnt + 1e-3)
    return dist, in_loss.to(torch.float32)


def proxy_voting_loss_v2(vertex_pred, keypoint_targets, vertex_one_hot_weights, vertex_weights, invert_weights=False, normalize=True, reduce=True,
                         loss_per_object=False, min_object_pixel=20):
    """loss_functions.py:132-203."""
    per_obj = bool(loss_per_object and normalize)
    img, osum, ocnt, _, elem, (b, h, w, ver_dim, oc, kp) = _proxy(vertex_pred, keypoint_targets, vertex_one_hot_weights, vertex_weights, invert_weights,
                                                                  per_obj, False, not normalize)
    if per_obj:
        cnt = ocnt.to(torch.float64)
        valid = (cnt >= min_object_pixel).to(torch.float64)
        obj = valid * osum / (ver_dim * cnt + 1e-3)
        n = valid.sum(1)
        in_loss = torch.where(n > 0, obj.sum(1) / torch.where(n > 0, n, torch.ones_like(n)), torch.zeros_like(n)).to(torch.float32)
    elif normalize:
        in_loss = (img[:, 0] / (ver_dim * img[:, 1] + 1e-3)).to(torch.float32)
    else:
        in_loss = elem
    return in_loss.mean() if reduce else in_loss


def keypoint_reprojection_loss(points_estimated, seg_estimated, poses_gt, object_points_3d, target_seg, camera_data, offsets, confidence,
                               max_pixel_error=25.0, confidence_regularization=False, points_gt=None, min_num=20, min_num_gt=-1,
                               use_bpnp_reprojection_loss=False, estimate_poses=False, filter_with_gt=True) -> Tuple[torch.Tensor, Optional[np.ndarray], torch.Tensor]:
    """loss_functions.py:207-344 -> (loss, poses_est or None, points_estimated in image pixels [B,oc,kp,2] (x,y), zero for unavailable objects).
    points_estimated [B,oc,kp,2] voted keypoints (y,x) in crop pixels; seg_estimated [B,H,W,K] logits; target_seg [B,H,W,K] one-hot;
    poses_gt [B,oc,ic,3,4]; object_points_3d [B,oc,ic,kp,3]; camera_data [B,3,3] (the first matrix is used, as in the reference); offsets
    [B,10]; confidence [B,H,W,kp] (only read with confidence_regularization)."""
    from ..train_engine import crop_to_image_affine, project_keypoints
    from ..training import _host, bpnp_reprojection_loss_host, poses_from_coords

    _need_cuda(seg_estimated)
    lib = _lib.load()
    dev = seg_estimated.device
    seg = _f32(seg_estimated)
    b, h, w, K = seg.shape
    oc = K - 1
    coords = _f32(points_estimated, dev).reshape(b, oc, -1, 2)
    kp = coords.shape[2]
    stream = _stream(seg)
    labels_gt = torch.argmax(target_seg.to(dev), dim=-1).to(torch.uint8).contiguous()
    labels_est = torch.empty(b, h, w, dtype=torch.uint8, device=dev)
    check(lib.cp_argmax_labels(seg.data_ptr(), K, K, b * h * w, labels_est.data_ptr(), stream), "cp_argmax_labels")
    conf = _f32(confidence, dev) if (confidence is not None and confidence_regularization) else torch.zeros(b, h, w, kp, dtype=torch.float32, device=dev)
    counts = torch.zeros(2, b, K, dtype=torch.int32, device=dev)
    conf_sums = torch.zeros(b, kp, dtype=torch.float64, device=dev)
    check(lib.cp_kp_stats_f32(conf.data_ptr(), conf.shape[-1], 0, labels_gt.data_ptr(), labels_est.data_ptr(), b, h, w, K, kp, counts.data_ptr(),
                              conf_sums.data_ptr(), stream), "cp_kp_stats_f32")
    avail = counts[1, :, 1:] > min_num
    if filter_with_gt:
        avail = avail & (counts[0, :, 1:] > (min_num if min_num_gt < 0 else min_num_gt))
    avail = avail.to(torch.float32).contiguous()
    cam = _host(camera_data)
    cam = cam[0] if cam.ndim == 3 else cam
    p3d = _host(object_points_3d).reshape(b, oc, -1, 3)
    gt_xy = torch.from_numpy(project_keypoints(p3d, cam, _host(poses_gt).reshape(b, oc, 3, 4))).to(dev).contiguous()
    aff = torch.from_numpy(crop_to_image_affine(_host(offsets))).to(dev).contiguous()
    batch = {"offsets": offsets, "cam_mat": camera_data, "keypoints3d": object_points_3d}
    poses_est = None
    if use_bpnp_reprojection_loss:
        lv, _, poses_est = bpnp_reprojection_loss_host(coords, gt_xy, aff, avail, p3d, cam, float(max_pixel_error), 1.0)
        loss = torch.tensor(lv, dtype=torch.float64, device=dev)
    else:
        val = torch.zeros(1, dtype=torch.float64, device=dev)
        g = torch.empty(b, oc, kp, 2, dtype=torch.float32, device=dev)
        check(lib.cp_kp_reproj_loss_f32(coords.data_ptr(), gt_xy.data_ptr(), aff.data_ptr(), avail.data_ptr(), b, oc, kp, float(max_pixel_error), 1.0,
                                        g.data_ptr(), val.data_ptr(), stream), "cp_kp_reproj_loss_f32")
        loss = val[0]
        if estimate_poses:
            poses_est, _ = poses_from_coords(coords, avail, batch)
    if confidence_regularization:
        cnt = counts[0, :, 1:].sum(dim=1, keepdim=True).double()
        safe = torch.where(cnt > 0, cnt, torch.ones_like(cnt))
        cl = torch.where(cnt > 0, conf_sums / safe, torch.zeros_like(conf_sums))
        loss = loss + torch.abs(cl - 0.7).mean()
    # points_estimated of the reference's return value: the voted keypoints mapped to image pixels (x,y), zero for unavailable objects
    A = aff.reshape(b, 1, 1, 2, 3)
    xy = coords.flip(-1)
    pts = torch.stack([A[..., 0, 0] * xy[..., 0] + A[..., 0, 1] * xy[..., 1] + A[..., 0, 2],
                       A[..., 1, 0] * xy[..., 0] + A[..., 1, 1] * xy[..., 1] + A[..., 1, 2]], dim=-1) * avail[:, :, None, None]
    return loss.to(torch.float32), poses_est, pts
