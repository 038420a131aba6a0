"""Pose recovery and ADD / ADD-S / 2-D projection evaluation on the host.

Counterparts of estimate_poses / evaluate_poses / map_estimates (casapose/pose_estimation/ransac_voting.py:487-687)
and of estimate_and_evaluate_poses / evaluate_pose_estimates / poses_pnp (pose_evaluation.py:11-217).  The reference runs
these through tf.map_fn + tf.numpy_function (PnP under the GIL); here they are plain NumPy fp64 over the tiny
[B, objects, kp] tensors the GPU voters hand back -- the PnP solve stays on the host by design (north_star).
Inputs may be torch tensors (any device) or arrays; outputs are NumPy float32 like the reference's.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

from . import pnp as _pnp


def _np(a, dtype=np.float64):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=dtype)


def transform_points_back(points_xy: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """transform_points_back_tf (ransac_voting.py:92-121) with the argument order of map_offsets (:494-504):
    crop pixels (x,y) -> original image pixels."""
    o = _np(offsets).reshape(10)
    hc, wc, dx, dy, ang, sc, sx, sy = o[0], o[1], o[4], o[5], o[6], o[7], o[8], o[9]
    p = _np(points_xy).reshape(-1, 2) / sc + np.array([wc, hc])
    p = p - np.array([dx, dy])
    ar = -ang * (np.pi / 180.0)
    a, b = np.cos(ar), np.sin(ar)
    cx, cy = sx / 2.0, sy / 2.0
    c = (1.0 - a) * cx - b * cy
    d = b * cx + (1.0 - a) * cy
    return np.stack([a * p[:, 0] + b * p[:, 1] + c, -b * p[:, 0] + a * p[:, 1] + d], axis=1)


def project(xyz: np.ndarray, K: np.ndarray, RT: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """project_tf (ransac_voting.py:173-182): -> (pixels [n,2] with divide_no_nan, camera-frame points [n,3])."""
    cam = xyz @ RT[:, :3].T + RT[:, 3:].T
    pix = cam @ K.T
    z = pix[:, 2:]
    return np.where(z != 0, pix[:, :2] / np.where(z != 0, z, 1.0), 0.0), cam


def estimate_poses(points, keypoints, camera_matrixes, valid_points_filter, offsets, rng=None):
    """points [b,oc,vn,2] (x,y) crop pixels; keypoints [b,oc,ic,vn,3]; camera_matrixes [b,3,3]; valid_points_filter [b,oc];
    offsets [b,10]  ->  (poses [b,oc,3,4], false_positive [oc]).  A point set summing to |.| < 0.01 means "object not
    voted" and yields the zero pose (map_offsets / map_pnp, :488-515); a non-empty vote for an object that is not in the
    ground truth counts as a false positive (:518-523)."""
    pts = _np(points)
    kp3 = _np(keypoints)
    cams = _np(camera_matrixes)
    valid = _np(valid_points_filter)
    offs = _np(offsets)
    b, oc, vn, _ = pts.shape
    poses = np.zeros((b, oc, 3, 4), np.float32)
    false_pos = np.zeros(oc, np.float32)
    for n in range(b):
        for o in range(oc):
            p = pts[n, o]
            if valid[n, o] == 0 and p.sum() > 0:
                false_pos[o] += 1.0
            if abs(p.sum()) < 0.01:
                continue
            p_img = transform_points_back(p, offs[n])
            cam = cams[n] if cams.ndim == 3 else cams
            poses[n, o] = _pnp.pnp(kp3[n, o, 0], p_img, cam, rng=rng)
    return poses, false_pos


def _adds_error(target: np.ndarray, est: np.ndarray) -> np.ndarray:
    """nearest-neighbour distances in fp64 (:596-611): sqrt(|min_j |a_i - b_j|^2| + 1e-5)."""
    try:
        from scipy.spatial import cKDTree

        d, _ = cKDTree(est).query(target, k=1)
        d2 = d * d
    except ImportError:  # pragma: no cover
        d2 = np.array([((est - a) ** 2).sum(axis=1).min() for a in target])
    return np.sqrt(np.abs(d2) + 1e-5)


SYMMETRIC_VERTEX_COUNTS = (7862, 3417)  # glue and eggbox meshes select ADD-S (ransac_voting.py:619)


def evaluate_poses(poses, poses_gt, points_estimated, object_points_3d, object_points_3d_count, camera_matrixes, diameters,
                   valid_points_filter, allowed_error_2d: float = 5.0):
    """-> (err_2d, err_3d, valid_2d, valid_3d, missing_object, valid_points_count, false_positive_detection), each [oc],
    summed over the batch (:627-687).  Per (image, object) (map_estimates, :561-624):
      not in GT: a non-zero pose is a false positive;  zero pose for a GT object: missing (errors 99.9 / 999.9);
      else mean 2-D reprojection distance and ADD (ADD-S when the mesh has 7862 or 3417 vertices), correct iff
      ADD < 0.1 * diameter and 2-D error < allowed_error_2d."""
    P, G = _np(poses), _np(poses_gt)
    pts3 = _np(object_points_3d)
    cnt = _np(object_points_3d_count, np.int64)
    cams, diam, valid = _np(camera_matrixes), _np(diameters), _np(valid_points_filter)
    b, oc = P.shape[0], P.shape[1]
    G = G.reshape(b, oc, -1, 3, 4)
    cnt = cnt.reshape(b, oc, -1)
    diam = diam.reshape(b, oc, -1)
    out = np.zeros((b, oc, 6), np.float32)
    for n in range(b):
        cam = cams[n] if cams.ndim == 3 else cams
        for o in range(oc):
            pose = P[n, o]
            if valid[n, o] == 0:
                out[n, o, 5] = 1.0 if abs(pose.sum()) > 1e-4 else 0.0
                continue
            if abs(pose.sum()) < 1e-4:
                out[n, o] = (99.9, 999.9, 0.0, 0.0, 1.0, 0.0)
                continue
            c = int(cnt[n, o, 0])
            X = pts3[n, o, 0][:c]
            p2, p3 = project(X, cam, pose)
            t2, t3 = project(X, cam, G[n, o, 0])
            e2 = np.linalg.norm(t2 - p2, axis=1).mean()
            if c in SYMMETRIC_VERTEX_COUNTS:
                e3 = _adds_error(t3, p3).mean()
            else:
                e3 = np.linalg.norm(t3 - p3, axis=1).mean()
            out[n, o] = (e2, e3, float(e3 < diam[n, o, 0] * 0.1), float(e2 < allowed_error_2d), 0.0, 0.0)
    s = out.sum(axis=0)
    valid_count = valid.sum(axis=0).astype(np.float32)
    return s[:, 0], s[:, 1], s[:, 3], s[:, 2], s[:, 4], valid_count, s[:, 5]


def _objects_available(target_seg, min_num: int) -> np.ndarray:
    seg = target_seg
    if hasattr(seg, "detach"):
        import torch

        if seg.dim() == 4:
            cnt = (seg[..., 1:] != 0).sum(dim=(1, 2))
        else:
            k = int(seg.max().item()) + 1
            cnt = torch.stack([(seg == c).sum(dim=(1, 2)) for c in range(1, k)], dim=1)
        return (cnt > min_num).cpu().numpy().astype(np.int32)
    seg = np.asarray(seg)
    return ((seg[..., 1:] != 0).sum(axis=(1, 2)) > min_num).astype(np.int32)


def _eval_points(object_points_3d, evaluation_points, object_points_3d_count, b, oc, ic):
    if evaluation_points is not None and object_points_3d_count is not None:
        ev = _np(evaluation_points)                       # [oc, V, 3]
        pts = np.tile(ev[None, :, None], (b, 1, ic, 1, 1))
        cnt = np.tile(_np(object_points_3d_count, np.int64).reshape(1, oc, -1)[:, :, :1], (b, 1, ic))
        return pts, cnt
    return _np(object_points_3d), np.full((b, oc, ic), 9, np.int64)


def evaluate_pose_estimates(points_estimated, poses, poses_gt, target_seg, object_points_3d, camera_data, diameters,
                            evaluation_points=None, object_points_3d_count=None, min_num: int = 20):
    """pose_evaluation.py:100-160: statistics for poses that were already estimated (the estimate_coords path of
    test_casapose.py:336-348).  -> ([valid_2d, valid_3d, valid_pose_count, zeros, err_2d, err_3d, missing, false_pos], poses,
    points_estimated)."""
    G = _np(poses_gt)
    b, oc, ic = G.shape[0], G.shape[1], G.shape[2]
    avail = _objects_available(target_seg, min_num)
    pts, cnt = _eval_points(object_points_3d, evaluation_points, object_points_3d_count, b, oc, ic)
    P = _np(poses).reshape(b, oc, 3, 4)
    e2, e3, v2, v3, miss, vcount, fp = evaluate_poses(P, G, points_estimated, pts, cnt, camera_data, diameters, avail, 5.0)
    return [v2, v3, vcount, np.zeros_like(v2), e2, e3, miss, fp], poses, points_estimated


def estimate_and_evaluate_poses(output_seg, target_seg, output_vertex, poses_gt, object_points_3d, camera_data, diameters, offsets,
                                evaluation_points=None, object_points_3d_count=None, points_estimated=None, min_num: int = 20,
                                draws=None):
    """pose_evaluation.py:11-97: RANSAC keypoint voting on the arg-max mask (unless points are given), host PnP, then
    evaluate_poses.  output_seg [b,h,w,K], output_vertex [b,h,w,2*kp] device tensors."""
    import torch

    from .ransac_voting import ransac_voting_layer_all_masks

    G = _np(poses_gt)
    b, oc, ic = G.shape[0], G.shape[1], G.shape[2]
    h, w = output_seg.shape[1], output_seg.shape[2]
    avail = _objects_available(target_seg, min_num)
    if points_estimated is None:
        lab = torch.argmax(output_seg, dim=3)
        onehot = torch.nn.functional.one_hot(lab, output_seg.shape[3])[..., 1:].to(torch.float32)
        kw = {} if draws is None else {"draws": draws}
        vc = _np(object_points_3d).shape[3]
        if oc > 1 and output_vertex.shape[-1] == vc * oc * 2:
            # `pvnet` with separated vector fields (pose_evaluation.py:38-45): every pixel votes with the slice of its own (arg-max) object,
            # background pixels with zeros -- an indexing step
            sl = output_vertex.reshape(b, h, w, oc, vc * 2)
            idx = torch.clamp(lab - 1, min=0)[..., None, None].expand(b, h, w, 1, vc * 2)
            output_vertex = torch.where((lab == 0)[..., None], torch.zeros((), dtype=sl.dtype, device=sl.device), torch.gather(sl, 3, idx)[:, :, :, 0])
        vert = output_vertex.reshape(b, h, w, -1, 2).contiguous()
        points_estimated = ransac_voting_layer_all_masks(onehot, vert, 512, inlier_thresh=0.99, max_iter=20, min_num=min_num, max_num=30000, **kw)
    else:
        points_estimated = _np(points_estimated) * np.array([[[[h, w]]]], np.float64)
    poses, false_positive_mask = estimate_poses(points_estimated, object_points_3d, camera_data, avail, offsets)
    pts, cnt = _eval_points(object_points_3d, evaluation_points, object_points_3d_count, b, oc, ic)
    e2, e3, v2, v3, miss, vcount, fp = evaluate_poses(poses, G, points_estimated, pts, cnt, camera_data, diameters, avail, 5.0)
    return [v2, v3, vcount, false_positive_mask, e2, e3, miss, fp], poses, points_estimated


def poses_pnp(points_estimated, seg_estimated, object_points_3d, camera_data, no_objects: int, min_num: int = 20, rng=None):
    """pose_evaluation.py:164-217: voted keypoints [b,oc,vc,2] in (y,x) -> poses [b,oc,1,3,4]; objects with <= min_num
    estimated pixels get the zero pose; the pose is negated when t_z < 0."""
    import torch

    pts = _np(points_estimated)[..., ::-1]  # tf.reverse: (y,x) -> (x,y)
    kp3 = _np(object_points_3d)
    b, oc = pts.shape[0], no_objects
    lab = torch.argmax(seg_estimated, dim=3) if hasattr(seg_estimated, "detach") else torch.from_numpy(np.argmax(seg_estimated, axis=3))
    cam = _np(camera_data)
    cam = cam[0] if cam.ndim == 3 else cam
    out = np.zeros((b, oc, 1, 3, 4), np.float32)
    for n in range(b):
        for o in range(oc):
            if int((lab[n] == o + 1).sum()) <= min_num:
                continue
            p6 = _pnp.pnp_rvec_t(kp3[n, o].reshape(-1, 3), pts[n, o], cam, rng=rng)
            if not np.all(np.isfinite(p6)):
                raise FloatingPointError("poses_pnp: non-finite pose for image %d object %d" % (n, o))
            R, t = _pnp.rodrigues(p6[:3]), p6[3:].astype(np.float64)
            P = np.concatenate([R, t.reshape(3, 1)], axis=1)
            out[n, o, 0] = -P if t[2] < 0 else P
    return out
