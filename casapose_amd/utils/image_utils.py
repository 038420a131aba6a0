"""Target vector fields with the reference's function names and argument conventions (casapose/utils/image_utils.py:17-79), computed by
cp_vector_field_f32.  Device tensors in, device tensors out; no CPU fallback.

    target_dirs = get_all_vectorfields(target_seg, target_vertex, filtered_seg, separated_vectorfields)      # train_casapose.py:528-533

`target_seg` [B,H,W,K] one-hot, `filtered_seg` [B,H,W,1] integer labels, `target_vertex` [B,oc,ic,kp,2] keypoints (y,x) in crop pixels --
the batch-tuple entries 1, 8 and 3 of the reference.
"""
from __future__ import annotations

import torch

from .. import _lib
from .._lib import check


def _labels_u8(mask: torch.Tensor) -> torch.Tensor:
    """[B,H,W,1] (or [B,H,W]) integer-valued mask -> contiguous uint8 label map (format adaptation only)."""
    if mask.dim() == 4:
        if mask.shape[-1] != 1:
            raise ValueError("mask must have one channel (got %s)" % (tuple(mask.shape),))
        mask = mask[..., 0]
    return mask.to(torch.uint8).contiguous()


def _field(labels: torch.Tensor, coords: torch.Tensor, separated: bool, normalize: bool) -> torch.Tensor:
    if not labels.is_cuda:
        raise _lib.CasaposeHipError("casapose.utils.image_utils needs CUDA (ROCm) tensors; there is no CPU fallback")
    b, h, w = labels.shape
    _, oc, ic, kp, _ = coords.shape
    k = coords.to(device=labels.device, dtype=torch.float32).contiguous()
    width = (oc if separated else 1) * 2 * kp
    out = torch.empty(b, h, w, width, dtype=torch.float32, device=labels.device)
    check(_lib.load().cp_vector_field_f32(labels.data_ptr(), k.data_ptr(), b, h, w, oc, ic, kp, 1 if separated else 0, 1 if normalize else 0, out.data_ptr(), width,
                                          torch.cuda.current_stream(labels.device).cuda_stream), "cp_vector_field_f32")
    return out


def compute_vertex_hcoords_batch_v3(mask: torch.Tensor, coords: torch.Tensor, use_motion: bool = False) -> torch.Tensor:
    """image_utils.py:17-63.  mask [B,H,W,1]: 0 = background, c = object class c; coords [B,classes,instances,points,2] (y,x).  Returns
    [B,H,W,points*2]: per foreground pixel the (l2-normalised unless use_motion) vectors from the pixel centre to the keypoints of its class --
    of the instance whose first keypoint is nearest when there are several -- and zeros on the background."""
    return _field(_labels_u8(mask), coords, False, not use_motion)


def get_all_vectorfields(target_seg: torch.Tensor, target_vertex: torch.Tensor, filtered_seg: torch.Tensor, separated_vectorfields: bool) -> torch.Tensor:
    """image_utils.py:66-79.  Merged field from `filtered_seg`, or -- the `pvnet` model -- one field per object from the object's channel of
    `target_seg`, concatenated ([B,H,W,oc*points*2]).  The separated form equals the merged kernel writing each pixel's vectors into the slot
    of its own class (the channels of a one-hot map are disjoint)."""
    if not separated_vectorfields:
        return compute_vertex_hcoords_batch_v3(filtered_seg, target_vertex)
    if target_seg.dim() != 4 or target_seg.shape[-1] != target_vertex.shape[1] + 1:
        raise ValueError("target_seg must be a [B,H,W,objects+1] one-hot map")
    fg = target_seg[..., 1:]
    labels = torch.where(fg.sum(-1) > 0, torch.argmax(fg, dim=-1) + 1, torch.zeros((), dtype=torch.int64, device=fg.device))
    return _field(labels.to(torch.uint8).contiguous(), target_vertex, True, True)
