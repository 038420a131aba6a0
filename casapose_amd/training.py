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

from typing import Callable, Dict, Optional, Sequence, Union

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


def train_step(net, batch: Dict[str, torch.Tensor], loss_factors, optimizer: Adam, opt, group=None, world_size: int = 1,
               train: bool = True):
    """One step on a batch dict with the reference's tuple fields (train_casapose.py:496-507):
       img [B,H,W,3]; target_seg [B,H,W,K] one-hot (or a uint8 label map); keypoints3d [B,oc,1,kp,3]; target_vert
       [B,oc,1,kp,2] 2-D keypoints (y,x) in crop pixels; cam_mat [B,3,3] or [3,3]; offsets [B,10]; filtered_seg
       (optional label map); poses_gt [B,oc,1,3,4].
    `opt` carries the config flags (train_vectors_with_ground_truth, estimate_coords, max_keypoint_pixel_error,
    confidence_regularization, use_bpnp_reprojection_loss).  Returns python floats
    [loss, mask_loss, vertex_loss, proxy_loss, kp_loss] (compute_loss, train_casapose.py:137-145)."""
    if getattr(opt, "use_bpnp_reprojection_loss", False):
        raise NotImplementedError("use_bpnp_reprojection_loss (BPnP backward, bpnp_layers.py:138-212) is not built yet")
    if getattr(loss_factors, "filter_high_proxy_errors", False):
        raise NotImplementedError("filter_high_proxy_errors (train_casapose.py:71-93) is not built yet")
    plan, dev = net.training_plan(batch["img"].shape[0], batch["img"].shape[1], batch["img"].shape[2], group, world_size)
    img = batch["img"].to(device=dev, dtype=torch.float32).contiguous()
    labels = labels_from_onehot(batch["target_seg"].to(dev))
    fg = labels_from_onehot(batch["filtered_seg"].to(dev)) if batch.get("filtered_seg") is not None else labels
    kpts = batch["target_vert"].to(device=dev, dtype=torch.float32)
    B, oc = kpts.shape[0], kpts.shape[1]
    kpts = kpts.reshape(B, oc, -1, 2).contiguous()
    cond = labels if getattr(opt, "train_vectors_with_ground_truth", False) else None
    plan.update_moving = train
    plan.forward(img, cond)
    wts = (float(loss_factors.mask_loss_weight), float(loss_factors.vertex_loss_weight), float(loss_factors.proxy_loss_weight))
    sums = plan.loss_and_grad(labels, fg, kpts, *wts, filter_with_segmentation=bool(loss_factors.filter_vertex_with_segmentation))
    kp_w = float(getattr(loss_factors, "kp_loss_weight", 0.0))
    kp_loss = None
    if getattr(opt, "estimate_coords", False):
        cam = np.asarray(batch["cam_mat"].cpu() if torch.is_tensor(batch["cam_mat"]) else batch["cam_mat"], np.float64)
        cam = cam[0] if cam.ndim == 3 else cam  # the reference uses camera_data[0] for the whole batch (loss_functions.py:322)
        p3d = np.asarray(batch["keypoints3d"].cpu() if torch.is_tensor(batch["keypoints3d"]) else batch["keypoints3d"], np.float64).reshape(B, oc, -1, 3)
        poses = np.asarray(batch["poses_gt"].cpu() if torch.is_tensor(batch["poses_gt"]) else batch["poses_gt"], np.float64).reshape(B, oc, 3, 4)
        offs = np.asarray(batch["offsets"].cpu() if torch.is_tensor(batch["offsets"]) else batch["offsets"], np.float64)
        gt_xy = torch.from_numpy(TE.project_keypoints(p3d, cam, poses)).to(dev).contiguous()
        aff = torch.from_numpy(TE.crop_to_image_affine(offs)).to(dev).contiguous()
        kp_loss = plan.kp_loss_and_grad(labels, gt_xy, aff, kp_w, max_pixel_error=float(getattr(opt, "max_keypoint_pixel_error", 25.0)), min_num=50,
                                        confidence_regularization=bool(getattr(opt, "confidence_regularization", False)) and train,
                                        vote_with_gt=bool(getattr(opt, "train_vectors_with_ground_truth", False)))
    if train:
        stream = torch.cuda.current_stream(dev).cuda_stream
        plan.backward()
        plan.all_reduce_grads()
        optimizer.apply(plan.store, stream)
        plan.refresh_weights(stream)
        net.mark_trained()
    s = sums.cpu().numpy()
    kpv = float(kp_loss.item()) if kp_loss is not None else 0.0
    total = wts[0] * s[0] + wts[1] * s[1] + wts[2] * s[2] + kp_w * kpv
    return [float(total), float(s[0]), float(s[1]), float(s[2]), kpv]
