"""Functional wrappers over the C ABI (one call = one or two kernel launches).

Every function takes contiguous float32 / uint8 CUDA tensors in NHWC and returns new device
tensors.  They are thin: argument checking lives in the library (cp_last_error), and
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check
from .engine import FusedConv


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CasaposeHipError("casapose_amd ops need CUDA (ROCm) tensors; there is no CPU fallback")


def conv2d_fused(sources: Sequence[torch.Tensor], kernel: np.ndarray, *, layout: int = 0, stride: int = 1, dilation: int = 1,
                 pad: int = 0, modes: Optional[Sequence[int]] = None, sels: Optional[Sequence[Optional[torch.Tensor]]] = None,
                 pre: Optional[Sequence[Optional[Tuple[torch.Tensor, torch.Tensor]]]] = None, real_channels: Optional[Sequence[int]] = None,
                 in_hw: Optional[Tuple[int, int]] = None, tap_label=None, row_scale=None, residual=None, scale=None, shift=None,
                 epi_label=None, act: int = 0, want_raw: bool = True, want_act: bool = False, tile_hint: int = 0):
    """One fused convolution (see cp_conv2d_fwd_f32 in include/casapose_hip.h).
    `kernel` is the Keras-layout host array (HWIO, or IHWO when layout=1).  Returns
    (out_raw | None, out_act | None)."""
    _need_cuda(*sources)
    dev = sources[0].device
    b = sources[0].shape[0]
    modes = list(modes) if modes is not None else [0] * len(sources)
    sels = list(sels) if sels is not None else [None] * len(sources)
    pre = list(pre) if pre is not None else [None] * len(sources)
    chans = [s.shape[3] for s in sources]
    real = list(real_channels) if real_channels is not None else chans
    kh, kw = (kernel.shape[0], kernel.shape[1]) if layout == 0 else (kernel.shape[1], kernel.shape[2])
    cout = kernel.shape[3]
    if in_hw is None:
        h0, w0 = sources[0].shape[1], sources[0].shape[2]
        in_hw = (h0, w0) if modes[0] == 0 else (2 * h0, 2 * w0)
    layer = FusedConv("conv2d_fused", kernel, layout, kh, kw, cout, list(zip(chans, real)), dev)
    eh, ew = (kh - 1) * dilation + 1, (kw - 1) * dilation + 1
    oh = (in_hw[0] + 2 * pad - eh) // stride + 1
    ow = (in_hw[1] + 2 * pad - ew) // stride + 1
    out_raw = torch.empty(b, oh, ow, cout, dtype=torch.float32, device=dev) if want_raw else None
    out_act = torch.empty(b, oh, ow, cout, dtype=torch.float32, device=dev) if want_act else None
    srcs = [dict(data=s, ld=s.shape[3], mode=m, sel=sl, pre=p) for s, m, sl, p in zip(sources, modes, sels, pre)]
    layer.bind(batch=b, in_h=in_hw[0], in_w=in_hw[1], stride=stride, dilation=dilation, pad=pad, srcs=srcs,
               tap_label=tap_label, row_scale=row_scale, residual=residual, scale=scale, shift=shift, epi_label=epi_label,
               act=act, out_raw=out_raw, out_act=out_act, tile_hint=tile_hint)
    layer.run(_stream(sources[0]))
    return out_raw, out_act


def pad_channels_3to4(img: torch.Tensor) -> torch.Tensor:
    _need_cuda(img)
    b, h, w, c = img.shape
    assert c == 3
    out = torch.empty(b, h, w, 4, dtype=torch.float32, device=img.device)
    check(_lib.load().cp_pad_channels_3to4(img.data_ptr(), out.data_ptr(), b * h * w, _stream(img)), "cp_pad_channels_3to4")
    return out


def maxpool3x3s2(x: torch.Tensor, scale=None, shift=None, relu=False) -> torch.Tensor:
    _need_cuda(x)
    b, h, w, c = x.shape
    out = torch.empty(b, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c, dtype=torch.float32, device=x.device)
    check(_lib.load().cp_maxpool3x3s2_f32(x.data_ptr(), b, h, w, c, scale.data_ptr() if scale is not None else None,
                                          shift.data_ptr() if shift is not None else None, int(relu), out.data_ptr(), _stream(x)),
          "cp_maxpool3x3s2_f32")
    return out


def upsample_bilinear_x2(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    b, h, w, c = x.shape
    out = torch.empty(b, 2 * h, 2 * w, c, dtype=torch.float32, device=x.device)
    check(_lib.load().cp_upsample_bilinear_x2_f32(x.data_ptr(), b, h, w, c, out.data_ptr(), _stream(x)), "cp_upsample_bilinear_x2_f32")
    return out


def guided_upsample_x2(x: torch.Tensor, sel: torch.Tensor) -> torch.Tensor:
    _need_cuda(x, sel)
    b, h, w, c = x.shape
    out = torch.empty(b, 2 * h, 2 * w, c, dtype=torch.float32, device=x.device)
    check(_lib.load().cp_guided_upsample_x2_f32(x.data_ptr(), sel.data_ptr(), b, h, w, c, out.data_ptr(), _stream(x)), "cp_guided_upsample_x2_f32")
    return out


def argmax_labels(logits: torch.Tensor, classes: Optional[int] = None, offset: int = 0) -> torch.Tensor:
    """uint8 arg-max over logits[..., offset:offset+classes]."""
    _need_cuda(logits)
    b, h, w, ld = logits.shape
    classes = classes if classes is not None else ld - offset
    out = torch.empty(b, h, w, dtype=torch.uint8, device=logits.device)
    check(_lib.load().cp_argmax_labels(logits.data_ptr() + 4 * offset, ld, classes, b * h * w, out.data_ptr(), _stream(logits)), "cp_argmax_labels")
    return out


def label_pyramid(labels0: torch.Tensor):
    """Returns (labels[4], pnorm[4], sel[3]) -- see cp_label_pyramid."""
    _need_cuda(labels0)
    b, h, w = labels0.shape
    dev = labels0.device
    hs = [h, h // 2, h // 4, h // 8]
    ws = [w, w // 2, w // 4, w // 8]
    labels = [labels0] + [torch.empty(b, hs[l], ws[l], dtype=torch.uint8, device=dev) for l in range(1, 4)]
    pnorm = [torch.empty(b, hs[l], ws[l], dtype=torch.float32, device=dev) for l in range(4)]
    # guided-upsampling selection maps exist only between levels whose sizes are exactly 2:1
    sel = [torch.empty(b, hs[l], ws[l], dtype=torch.uint8, device=dev) if (hs[l] == 2 * hs[l + 1] and ws[l] == 2 * ws[l + 1] and hs[l + 1] > 0) else None
           for l in range(3)]
    lab = (C.c_void_p * 4)(*[t.data_ptr() if t.numel() else None for t in labels])
    pn = (C.c_void_p * 4)(*[t.data_ptr() if t.numel() else None for t in pnorm])
    sl = (C.c_void_p * 3)(*[t.data_ptr() if t is not None else None for t in sel])
    check(_lib.load().cp_label_pyramid(labels0.data_ptr(), b, h, w, lab, pn, sl, _stream(labels0)), "cp_label_pyramid")
    return labels, pnorm, sel


def ls_vote(field: torch.Tensor, seg_off: int, dir_off: int, conf_off: int, objects: int, kp: int = 9,
            labels: Optional[torch.Tensor] = None, return_sums: bool = False, sigmoid_weights: bool = False):
    """cp_ls_vote_f32 on a [B,H,W,ld] record tensor.  Returns keypoints [B,objects,kp,2] (y,x)."""
    _need_cuda(field, labels)
    b, h, w, ld = field.shape
    lib = _lib.load()
    sums = torch.empty(b, objects, kp, 5, dtype=torch.float64, device=field.device)
    out = torch.empty(b, objects, kp, 2, dtype=torch.float32, device=field.device)
    check(lib.cp_ls_vote_w_f32(field.data_ptr(), ld, seg_off, dir_off, conf_off, labels.data_ptr() if labels is not None else None,
                               b, h, w, objects, kp, 1 if sigmoid_weights else 0, sums.data_ptr(), out.data_ptr(), _stream(field)), "cp_ls_vote_w_f32")
    return (out, sums) if return_sums else out
