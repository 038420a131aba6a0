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
    base = seg._base if seg._base is not None else None
    same = base is not None and all(t._base is base for t in ts) and base.dim() == 4 and base.is_contiguous()
    if same:
        ld = base.shape[3]
        ok = all(t.stride() == (base.stride(0), base.stride(1), base.stride(2), 1) for t in ts)
        offs = [t.storage_offset() - base.storage_offset() for t in ts]
        if ok and ld % 4 == 0 and ld <= 64 and all(0 <= o < ld for o in offs):
            return base, offs
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
        if sigmoid_weights:
            raise NotImplementedError("sigmoid_weights=True (voting_layers_2d.py:32-33) is not built; the configs use softplus")
        if output_second_largest_component:
            raise NotImplementedError("output_second_largest_component is a reference debugging switch and is not built")
        self.name = name
        self.num_classes = num_classes
        self.num_points = num_points
        self.filter_estimates = filter_estimates
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
            lab0 = ops.argmax_labels(rec, classes=objects + 1, offset=so)
            ws = torch.empty(lib.cp_ccl_workspace_bytes(b, h, w, objects), dtype=torch.uint8, device=rec.device)
            labels = torch.empty_like(lab0)
            check(lib.cp_ccl_filter_labels(lab0.data_ptr(), b, h, w, objects, self.min_component, ws.data_ptr(),
                                           labels.data_ptr(), torch.cuda.current_stream(rec.device).cuda_stream),
                  "cp_ccl_filter_labels")
        return ops.ls_vote(rec, so, do, co, objects, self.num_points, labels=labels)
