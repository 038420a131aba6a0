"""CoordLSVotingWeighted with the reference's constructor / call convention
(casapose/pose_estimation/voting_layers_2d.py:5-122), computed by cp_ls_vote_f32
(+ cp_ccl_filter_labels when filter_estimates=True).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import _lib, ops
from .._lib import check


def _as_record(seg: torch.Tensor, direct: torch.Tensor, conf: torch.Tensor):
    """If the three inputs are channel slices of ONE [B,H,W,ld] tensor (the usual
    `tf.split(output_net, ...)` of train_casapose.py:539) use it in place; otherwise pack them."""
    ts = (seg, direct, conf)
    b, h, w = seg.shape[:3]
    ld = seg.stride(2)
    want = (h * w * ld, w * ld, ld, 1)
    same_store = all(t.untyped_storage().data_ptr() == seg.untyped_storage().data_ptr() for t in ts)
    if same_store and all(t.dtype == torch.float32 and t.stride() == want for t in ts) and ld % 4 == 0 and 0 < ld <= 64:
        first = min(t.storage_offset() for t in ts)
        offs = [t.storage_offset() - first for t in ts]
        fits = first + b * h * w * ld <= seg.untyped_storage().nbytes() // 4
        if fits and all(o + t.shape[3] <= ld for o, t in zip(offs, ts)) and (seg.untyped_storage().data_ptr() + 4 * first) % 16 == 0:
            return torch.as_strided(seg, (b, h, w, ld), want, storage_offset=first), offs
    k, d, c = seg.shape[3], direct.shape[3], conf.shape[3]
    ld = (k + d + c + 3) // 4 * 4
    rec = torch.zeros(seg.shape[0], seg.shape[1], seg.shape[2], ld, dtype=torch.float32, device=seg.device)
    rec[..., :k] = seg
    rec[..., k:k + d] = direct
    rec[..., k + d:k + d + c] = conf
    return rec, [0, k, k + d]


class CoordLSVotingWeighted:
    def __init__(self, name, num_classes, num_points=9, sigmoid_weights=False, filter_estimates=False,
                 output_second_largest_component=False):
        self.name = name
        # voting_layers_2d.py:58-59,71-73 ("just for testing"): vote with the SECOND largest component of each object (three histogram bins,
        # the third entry of top_k) instead of the largest; only read with filter_estimates
        self.output_second_largest_component = bool(output_second_largest_component)
        self.num_classes = num_classes
        self.num_points = num_points
        self.filter_estimates = filter_estimates
        self.sigmoid_weights = bool(sigmoid_weights)   # w = sigmoid(sigmoid_scale * conf), sigmoid_scale = 1 (voting_layers_2d.py:20,32-33); default softplus
        self.min_component = 50  # voting_layers_2d.py:66

    def __call__(self, inp: Sequence[torch.Tensor], **kwargs) -> torch.Tensor:
        seg, direct, conf = inp
        if not seg.is_cuda:
            raise _lib.CasaposeHipError("CoordLSVotingWeighted needs CUDA (ROCm) tensors; there is no CPU fallback")
        rec, (so, do, co) = _as_record(seg, direct, conf)
        objects = seg.shape[3] - 1
        labels: Optional[torch.Tensor] = None
        if self.filter_estimates:
            lib = _lib.load()
            b, h, w, _ = rec.shape
            from ..engine import cached_labels

            lab0 = None
            if so == 0 and rec.storage_offset() == 0:
                lab0 = cached_labels(rec.untyped_storage().data_ptr(), (b, h, w), seg._version)  # the forward's own arg-max map
            if lab0 is None:
                lab0 = ops.argmax_labels(rec, classes=objects + 1, offset=so)
            ws = torch.empty(lib.cp_ccl_workspace_bytes(b, h, w, objects), dtype=torch.uint8, device=rec.device)
            labels = torch.empty_like(lab0)
            check(lib.cp_ccl_filter_labels(lab0.data_ptr(), b, h, w, objects, self.min_component, 2 if self.output_second_largest_component else 1, ws.data_ptr(),
                                           labels.data_ptr(), torch.cuda.current_stream(rec.device).cuda_stream),
                  "cp_ccl_filter_labels")
        return ops.ls_vote(rec, so, do, co, objects, self.num_points, labels=labels, sigmoid_weights=self.sigmoid_weights)
