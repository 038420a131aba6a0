"""TEST INFRASTRUCTURE -- fp64 PyTorch restatement of the reference's loss and target-field FUNCTIONS with their own argument lists
(casapose/utils/loss_functions.py:14-203, casapose/utils/image_utils.py:17-79, the separated-field branch of compute_loss,
train_casapose.py:40-145).  Parity unpinned (TensorFlow is absent here); only tests/ may import this module.  tf.gather / gather_nd with
batch_dims, unsorted_segment_sum and divide_no_nan are written out with plain indexing.
"""
from __future__ import annotations

import torch


def _sl1(a):
    return torch.where(a < 1.0, 0.5 * a * a, a - 0.5)


def _weights(w, ignore, invert):
    if ignore:
        return torch.ones_like(w)
    return (1.0 - w).abs() if invert else w


def smooth_l1_loss(pred, target, weights, ignore_weights=False, invert_weights=False, normalize=True, reduce=True):
    """loss_functions.py:14-44"""
    b, h, w, c = pred.shape
    wt = _weights(weights, ignore_weights, invert_weights)
    v = _sl1((wt * (pred - target)).abs())
    if normalize:
        v = v.reshape(b, -1).sum(1) / (c * wt.reshape(b, -1).sum(1) + 1e-3)
    return v.mean() if reduce else v


def _perp_dist(pred, keypoint_targets, one_hot):
    """[b,h,w,kp] minimum over instances of |v_y (k_x - c_x) - v_x (k_y - c_y)| / |v| (0 where |v| = 0), keypoints of argmax(one_hot)
    (loss_functions.py:90-112 / 151-173: the reference writes it with keypoints rotated to (k_y, -k_x) and coordinates (-c_x, c_y))."""
    b, h, w, c = pred.shape
    kp = c // 2
    idx = torch.argmax(one_hot, dim=-1)                                   # [b,h,w]; all-zero rows -> 0
    bi = torch.arange(b)[:, None, None]
    k = keypoint_targets[bi, idx]                                         # [b,h,w,ic,kp,2] (y,x)
    v = pred.reshape(b, h, w, kp, 2)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=pred.dtype) + 0.5, torch.arange(w, dtype=pred.dtype) + 0.5, indexing="ij")
    num = (v[..., None, :, 0] * (k[..., 1] - xx[None, :, :, None, None]) - v[..., None, :, 1] * (k[..., 0] - yy[None, :, :, None, None])).abs()   # [b,h,w,ic,kp]
    nrm = (v * v).sum(-1).sqrt()[..., None, :]
    d = torch.where(nrm > 0, num / torch.where(nrm > 0, nrm, torch.ones_like(nrm)), torch.zeros_like(num))
    return d.min(dim=3).values, idx


def _segment_sums(values, idx, count):
    """tf.math.unsorted_segment_sum per image: values [b,h,w], idx [b,h,w] -> [b,count]"""
    b = values.shape[0]
    out = torch.zeros(b, count, dtype=values.dtype)
    out.scatter_add_(1, idx.reshape(b, -1), values.reshape(b, -1))
    return out


def proxy_voting_dist(pred, keypoint_targets, one_hot, weights, invert_weights=False, min_object_pixel=20):
    """loss_functions.py:47-129 -> (dist [b,h,w,kp], per-object loss [b,oc])"""
    b, h, w, ver_dim = pred.shape
    oc, kp = one_hot.shape[-1], keypoint_targets.shape[3]
    if oc > 1 and ver_dim == oc * kp * 2:                                 # separated field: the slice of the pixel's own object
        sl = pred.reshape(b, h, w, oc, kp * 2)
        idx = torch.argmax(one_hot, dim=3)
        sl = sl[torch.arange(b)[:, None, None], torch.arange(h)[None, :, None], torch.arange(w)[None, None, :], idx]
        pred = torch.where(weights > 0, torch.zeros_like(sl), sl)
        ver_dim = pred.shape[-1]
    wt = _weights(weights, False, invert_weights)
    d, idx = _perp_dist(pred, keypoint_targets, one_hot)
    dist = (wt * d).abs()
    mask_sum = one_hot.sum((1, 2))
    valid = (mask_sum >= min_object_pixel).to(pred.dtype)
    seg = _segment_sums(_sl1(dist).sum(-1), idx, oc)
    return dist, valid * seg / ((ver_dim / 2) * mask_sum + 1e-3)


def proxy_voting_loss_v2(pred, keypoint_targets, one_hot, weights, invert_weights=False, normalize=True, reduce=True, loss_per_object=False,
                         min_object_pixel=20):
    """loss_functions.py:132-203"""
    b, h, w, ver_dim = pred.shape
    oc = one_hot.shape[-1]
    wt = _weights(weights, False, invert_weights)
    d, idx = _perp_dist(pred, keypoint_targets, one_hot)
    dist = (wt * d).abs()
    if loss_per_object and normalize:
        mask_sum = one_hot.sum((1, 2))
        valid = (mask_sum >= min_object_pixel).to(pred.dtype)
        obj = valid * _segment_sums(_sl1(dist).sum(-1), idx, oc) / (ver_dim * mask_sum + 1e-3)
        n = valid.sum(1)
        v = torch.where(n > 0, obj.sum(1) / torch.where(n > 0, n, torch.ones_like(n)), torch.zeros_like(n))
    else:
        v = _sl1(dist)
        if normalize:
            v = v.reshape(b, -1).sum(1) / (ver_dim * wt.reshape(b, -1).sum(1) + 1e-3)
    return v.mean() if reduce else v


def compute_vertex_hcoords_batch_v3(mask, coords, use_motion=False):
    """image_utils.py:17-63.  mask [b,h,w,1] integer classes (0 = background), coords [b,classes,instances,points,2] (y,x)."""
    b, h, w = mask.shape[:3]
    _, c, o, m, _ = coords.shape
    cz = torch.cat([torch.zeros(b, 1, o, m, 2, dtype=coords.dtype), coords], dim=1)
    lab = mask[..., 0].to(torch.int64)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=coords.dtype) + 0.5, torch.arange(w, dtype=coords.dtype) + 0.5, indexing="ij")
    grid = torch.stack([yy, xx], -1)[None]                                 # [1,h,w,2]
    on_mask = cz[torch.arange(b)[:, None, None], lab]                      # [b,h,w,o,m,2]
    if o > 1:
        centers = on_mask[:, :, :, :, 0, :]                                # [b,h,w,o,2]
        nearest = torch.where(lab == 0, torch.zeros_like(lab), (grid[:, :, :, None, :] - centers).norm(dim=-1).argmin(dim=-1))
        tgt = on_mask[torch.arange(b)[:, None, None], torch.arange(h)[None, :, None], torch.arange(w)[None, None, :], nearest]
    else:
        tgt = on_mask[:, :, :, 0]
    dirs = (tgt - grid[:, :, :, None, :]) * (lab != 0)[..., None, None].to(coords.dtype)
    if not use_motion:
        dirs = dirs * torch.rsqrt(torch.clamp((dirs * dirs).sum(-1, keepdim=True), min=1e-12))     # tf.math.l2_normalize
    return dirs.reshape(b, h, w, m * 2)


def get_all_vectorfields(target_seg, target_vertex, filtered_seg, separated_vectorfields):
    """image_utils.py:66-79"""
    if not separated_vectorfields:
        return compute_vertex_hcoords_batch_v3(filtered_seg, target_vertex)
    parts = [compute_vertex_hcoords_batch_v3(target_seg[..., i + 1:i + 2], target_vertex[:, i:i + 1]) for i in range(target_seg.shape[3] - 1)]
    return torch.cat(parts, dim=3)


def compute_loss_separated(output_seg, target_seg, output_vert, target_vert, target_points, filter_vertex_with_segmentation=False):
    """compute_loss for separated vector fields (train_casapose.py:40-145 with separated_vectors): (mask, vertex, proxy).  target_seg is the
    one-hot the losses weight with (filtered_seg already applied by the caller); the arg-max filter of :64-69 is applied here."""
    oc = target_seg.shape[3] - 1
    vc = target_points.shape[3] * 2
    logp = torch.log_softmax(output_seg, dim=-1)
    mask_loss = -(target_seg * logp).sum(-1).mean()
    if filter_vertex_with_segmentation:
        same = (target_seg.argmax(-1) == output_seg.argmax(-1))[..., None]
        bg = torch.zeros_like(target_seg)
        bg[..., 0] = 1.0
        target_seg = torch.where(same, target_seg, bg)
    target_seg = target_seg.detach()
    vertex = sum(smooth_l1_loss(output_vert[..., i * vc:(i + 1) * vc], target_vert[..., i * vc:(i + 1) * vc], target_seg[..., i + 1:i + 2]) for i in range(oc))
    proxy = sum(proxy_voting_loss_v2(output_vert[..., i * vc:(i + 1) * vc], target_points[:, i:i + 1], one_hot=target_seg[..., i + 1:i + 2],
                                     weights=target_seg[..., i + 1:i + 2]) for i in range(oc))
    return mask_loss, vertex, proxy
