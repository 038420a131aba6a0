"""Gradient oracle for the training path (TEST INFRASTRUCTURE ONLY).

A PyTorch-CPU (fp64) restatement of casapose_c_gcu5 in TRAINING mode and of the losses, written
with differentiable torch ops so that autograd provides reference gradients for the HIP backward
kernels.  Same status as casapose_oracle.py: parity with the TensorFlow reference is UNPINNED (it
cannot be imported here); the forward of this file is cross-checked against the NumPy oracle in
tests/test_train_oracle.py, and every function cites the reference lines it restates.

Only tests/ and smoke checks may import this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 2e-5
BN_MOMENTUM = 0.99  # resnet.py:43 -- moving = 0.99*moving + 0.01*batch

STAGE_FILTERS = (64, 128, 256, 512)
STAGE_STRIDE = (1, 2, 1, 1)
STAGE_DILATION = (1, 1, 2, 4)


def to_torch(params: Dict[str, np.ndarray], dtype=torch.float64, requires_grad=True) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        trainable = not k.endswith(("moving_mean", "moving_variance"))
        out[k] = t.requires_grad_(requires_grad and trainable)
    return out


def conv_nhwc(x, w_hwio, stride=1, dilation=1, pad=0):
    """layers.Conv2D (valid after explicit zero padding), NHWC / HWIO."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w_hwio.permute(3, 2, 0, 1), stride=stride, dilation=dilation, padding=pad)
    return y.permute(0, 2, 3, 1)


def batchnorm_train(x, gamma, beta, stats_out: Optional[dict] = None, name: str = ""):
    """(Sync)BatchNormalization, training=True: normalise with the BIASED batch variance over
    (N,H,W); the moving statistics are updated with the same biased variance."""
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    if stats_out is not None:
        stats_out[name] = (mean.detach(), var.detach())
    y = (x - mean) / torch.sqrt(var + BN_EPS)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y


def leaky_pair(x):
    return F.relu(x) - F.relu(-0.1 * x)  # casapose.py:98-105


def maxpool_zero_pad(x):
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))  # zero padding (resnet.py:253)
    return F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)


def bilinear_x2(x):
    return F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)


def labels_pyramid(labels: torch.Tensor) -> List[torch.Tensor]:
    out = [labels]
    for _ in range(3):
        out.append(out[-1][:, ::2, ::2])
    return out


def partial_conv(x, w_ihwo, labels):
    """PartialConvolution with a hard label map: m(p,n) = [label(p+n) == label(p)] inside the image
    (_normalization_layers.py:333-371)."""
    b, h, w, c = x.shape
    w_hwio = w_ihwo.permute(1, 2, 0, 3)
    lab = labels.to(torch.int64)
    lp = F.pad(lab + 1, (1, 1, 1, 1))  # 0 = outside
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    out = torch.zeros(b, h, w, w_ihwo.shape[3], dtype=x.dtype)
    cnt = torch.zeros(b, h, w, dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            m = (lp[:, ky : ky + h, kx : kx + w] == (lab + 1)).to(x.dtype)
            cnt = cnt + m
            out = out + (xp[:, ky : ky + h, kx : kx + w, :] * m[..., None]) @ w_hwio[ky, kx]
    return out * (9.0 / cnt)[..., None]


def clade_train(x, labels, gamma_kc, beta_kc, stats_out=None, name=""):
    """ClassAdaptiveWeightedNormalization, training=True, hard labels (_normalization_layers.py:119-139)."""
    xn = batchnorm_train(x, None, None, stats_out, name)
    lab = labels.to(torch.int64)
    return gamma_kc[lab] * xn + beta_kc[lab]


def guided_select(lab_lo: torch.Tensor, lab_hi: torch.Tensor) -> torch.Tensor:
    """neighbour index 0..3 per hi-res pixel (_normalization_layers.py:534-551); labels are int maps."""
    b, h2, w2 = lab_lo.shape
    lp = F.pad(lab_lo + 1, (0, 1, 0, 1))  # bottom/right zero pad (label 0 never matches label+1 >= 1)
    cands = torch.stack([lp[:, :h2, :w2], lp[:, :h2, 1:], lp[:, 1:, :w2], lp[:, 1:, 1:]], dim=-1)
    cu = cands.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    eq = cu == (lab_hi + 1)[..., None]
    score = eq.to(torch.int64) * torch.tensor([4, 3, 2, 1])
    return torch.argmax(score, dim=-1)  # all-zero -> 0


def guided_upsample(x, lab_lo, lab_hi):
    b, h2, w2, c = x.shape
    sel = guided_select(lab_lo.to(torch.int64), lab_hi.to(torch.int64))
    yy, xx = torch.meshgrid(torch.arange(2 * h2), torch.arange(2 * w2), indexing="ij")
    sy = torch.clamp(yy[None] // 2 + sel // 2, max=h2 - 1)
    sx = torch.clamp(xx[None] // 2 + sel % 2, max=w2 - 1)
    bi = torch.arange(b)[:, None, None]
    return x[bi, sy, sx, :]


def guided_bilinear_upsample(x, lab_lo, lab_hi):
    """GuidedBilinearUpsampling with hard labels (_normalization_layers.py:607-664): taps whose low-res label differs from the
    hi-res label are replaced by the mean of the matching taps; fixed sub-pixel weights."""
    b, h2, w2, c = x.shape
    lo = lab_lo.to(torch.int64) + 1
    lp = F.pad(lo, (0, 1, 0, 1))
    cands = torch.stack([lp[:, :h2, :w2], lp[:, :h2, 1:], lp[:, 1:, :w2], lp[:, 1:, 1:]], dim=-1)
    cu = cands.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    cond = (cu == (lab_hi.to(torch.int64) + 1)[..., None]).to(x.dtype)          # [b,H,W,4]
    xp = F.pad(x, (0, 0, 0, 1, 0, 1))
    taps = torch.stack([xp[:, :h2, :w2], xp[:, :h2, 1:], xp[:, 1:, :w2], xp[:, 1:, 1:]], dim=3)
    taps = taps.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)         # [b,H,W,4,c]
    n = cond.sum(-1, keepdim=True)
    mean = (taps * cond[..., None]).sum(3, keepdim=True) / torch.clamp(n, min=1.0)[..., None]
    filled = torch.where(cond[..., None] > 0, taps, mean * (n[..., None] > 0))
    interp = torch.tensor([[1.0, 0, 0, 0], [0.5, 0.5, 0, 0], [0.5, 0, 0.5, 0], [0.25, 0.25, 0.25, 0.25]], dtype=x.dtype)
    yy, xx = torch.meshgrid(torch.arange(2 * h2), torch.arange(2 * w2), indexing="ij")
    wts = interp[(yy % 2) * 2 + (xx % 2)]                                        # [H,W,4]
    return (filled * wts[None, :, :, :, None]).sum(3)


def batchnorm_inference(x, gamma, beta, mean, var):
    """(Sync)BatchNormalization, training=False: the moving statistics (resnet.py:39-49, eps 2e-5)."""
    y = (x - mean) / torch.sqrt(var + BN_EPS)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y


def forward_train(p: Dict[str, torch.Tensor], img: torch.Tensor, labels: Optional[torch.Tensor], stats_out: Optional[dict] = None,
                  partial=(True,) * 5, guided=(False, True, True, True, False), bilinear=(False,) * 5, pvnet: bool = False,
                  shared=(False,) * 5, reuse_first: bool = False, skips2: bool = True, training: bool = True,
                  act_pattern: Optional[dict] = None, preact_out: Optional[dict] = None):
    """casapose_c_gcu5 (or a sibling: per decoder-2 block `partial` convolution / `guided` upsampling flags, else an ordinary
    convolution / plain nearest upsampling; pose_models.py:14-635) with training=True and decoder 2 conditioned on the given
    hard label map (the `data_segmentation` input of config_8.ini:71; pose_models.py:550-554).  Returns [B,H,W,K+ver_dim].
    training=False normalises with the moving statistics (the inference graph); labels=None conditions decoder 2 on the arg-max of
    the network's own logits (the estimated mask of pose_models.py:548-549, README.md:74-80).  That combination, in fp32 on all
    host cores, is the CPU baseline `bench.py` times beside the GPU.

    Kink analysis of the gradient tests: `preact_out` (dict) receives the input of every ReLU / leaky pair under the normalisation
    layer's name; `act_pattern` {layer name: bool tensor} REPLACES the branch decision z > 0 of those activations by a given pattern
    (relu(z) -> z*m, leaky(z) -> z*(m + 0.1(1-m))).  A device forward in fp32 takes the other branch at the few elements whose
    pre-activation is within rounding of zero; the gradient it must then produce is the fp64 gradient of the network WITH ITS pattern,
    and the forward value changes by at most 1.1|z| at those elements."""

    def act(name, z, leaky):
        if preact_out is not None:
            preact_out[name] = z.detach()
        if act_pattern is not None and name in act_pattern:
            m = act_pattern[name].to(z.dtype)
            return z * (m + 0.1 * (1.0 - m)) if leaky else z * m
        return leaky_pair(z) if leaky else F.relu(z)

    def bn(name, x):
        if not training:
            return batchnorm_inference(x, p.get(name + ".gamma"), p.get(name + ".beta"), p[name + ".moving_mean"], p[name + ".moving_variance"])
        return batchnorm_train(x, p.get(name + ".gamma"), p.get(name + ".beta"), stats_out, name)

    x = bn("bn_data", img)
    x = conv_nhwc(x, p["conv0.kernel"], stride=2, pad=3)
    x2s = act("bn0", bn("bn0", x), False)
    x = maxpool_zero_pad(x2s)
    taps = []
    for s in range(4):
        d = STAGE_DILATION[s]
        for u in range(2):
            base = "stage%d_unit%d_" % (s + 1, u + 1)
            stride = STAGE_STRIDE[s] if u == 0 else 1
            a = act(base + "bn1", bn(base + "bn1", x), False)
            shortcut = conv_nhwc(a, p[base + "sc.kernel"], stride=stride) if u == 0 else x
            y = conv_nhwc(a, p[base + "conv1.kernel"], stride=stride, dilation=d, pad=d)
            y = act(base + "bn2", bn(base + "bn2", y), False)
            y = conv_nhwc(y, p[base + "conv2.kernel"], dilation=d, pad=d)
            x = y + shortcut
            if u == 0 and s > 0:
                taps.append(a)
    x32s = act("bn1", bn("bn1", x), False)
    x4s, x8s, _x16s = taps
    skips = [None, x8s, x4s, x2s, img]
    d1 = None
    for i in range(5):
        n = "pv_block_%d" % (i + 1)
        inp = x32s if i == 0 else torch.cat([d1, skips[i]], dim=3)
        if shared[i]:  # one-input PartialConvolution pv_block_{i+1}_{i+6}_conv2d: ordinary SAME conv, [Cin,3,3,Cout] weights
            y = conv_nhwc(inp, p["pv_block_%d_%d_conv2d.weights" % (i + 1, i + 6)].permute(1, 2, 0, 3), pad=1)
        else:
            y = conv_nhwc(inp, p[n + "_conv2d.kernel"], pad=1)
        if i == 0:
            y_raw = y
        y = act(n + "_bn", bn(n + "_bn", y), i != 0)
        if 0 < i < 4:
            y = bilinear_x2(y)
        d1 = y
    if pvnet:  # PVNet: one merged head (pose_models.py:678)
        return conv_nhwc(d1, p["pv_final_conv.kernel"])
    logits = conv_nhwc(d1, p["pv_final_conv_segmentation.kernel"])
    if labels is None:
        labels = torch.argmax(logits.detach(), dim=-1)  # softmax(1e6 * logits) is one-hot at the first maximum
    labs = labels_pyramid(labels)
    lvl = [3, 3, 2, 1, 0]
    d2 = None
    for i in range(5):
        n = "pv_block_%d" % (i + 6)
        inp = x32s if i == 0 else (torch.cat([d2, skips[i]], dim=3) if skips2 else d2)
        lab = labs[lvl[i]]
        if i == 0 and reuse_first:  # casa_layer(y, "6", skip_conv=True) on the raw output of block 1's convolution
            y = y_raw
        elif shared[i]:
            w = p["pv_block_%d_%d_conv2d.weights" % (i + 1, i + 6)]
            y = partial_conv(inp, w, lab) if partial[i] else conv_nhwc(inp, w.permute(1, 2, 0, 3), pad=1)
        elif partial[i]:
            y = partial_conv(inp, p[n + "_prepare_conv2d.weights"], lab)
        else:
            y = conv_nhwc(inp, p[n + "_conv2d.kernel"], pad=1)
        if training:
            y = clade_train(y, lab, p[n + "_clade.gamma"], p[n + "_clade.beta"], stats_out, n + "_clade")
        else:
            y = batchnorm_inference(y, None, None, p[n + "_clade.moving_mean"], p[n + "_clade.moving_variance"])
            y = p[n + "_clade.gamma"][lab.to(torch.int64)] * y + p[n + "_clade.beta"][lab.to(torch.int64)]
        y = act(n + "_clade", y, i != 0)
        if 0 < i < 4:
            if guided[i] and bilinear[i]:
                y = guided_bilinear_upsample(y, lab, labs[lvl[i] - 1])
            elif guided[i]:
                y = guided_upsample(y, lab, labs[lvl[i] - 1])
            else:
                y = y.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)  # UpSampling2D(nearest), casapose.py:126-131
        d2 = y
    vertex = conv_nhwc(d2, p["pv_final_conv_vertex.kernel"])
    return torch.cat([logits, vertex], dim=3)


def kink_report(pattern: Dict[str, torch.Tensor], preact: Dict[str, torch.Tensor]):
    """(elements whose branch in `pattern` differs from the sign of the fp64 pre-activation, elements in all, largest |pre-activation| among the
    differing ones).  A device forward in fp32 may only take another branch where the fp64 pre-activation is within its rounding error of
    zero; the gradient tests assert that margin and then compare gradients against forward_train(act_pattern=pattern)."""
    flips, total, margin = 0, 0, 0.0
    for name, m in pattern.items():
        z = preact[name]
        d = (z > 0) != m.to(torch.bool)
        flips += int(d.sum())
        total += m.numel()
        if d.any():
            margin = max(margin, float(z[d].abs().max()))
    return flips, total, margin


# --------------------------------------------------------------------------------------
#  targets and losses (train_casapose.py:40-145; utils/loss_functions.py; utils/image_utils.py)
# --------------------------------------------------------------------------------------


def target_vector_field(labels: torch.Tensor, keypoints_yx: torch.Tensor) -> torch.Tensor:
    """compute_vertex_hcoords_batch_v3 for one instance per object (image_utils.py:17-63):
    unit vectors from each foreground pixel centre to its object's keypoints, (dy,dx) pairs, zero on
    background.  labels [B,H,W] int, keypoints [B,oc,kp,2] in (y,x)."""
    b, h, w = labels.shape
    kp = keypoints_yx.shape[2]
    yy, xx = torch.meshgrid(torch.arange(h, dtype=keypoints_yx.dtype) + 0.5, torch.arange(w, dtype=keypoints_yx.dtype) + 0.5, indexing="ij")
    grid = torch.stack([yy, xx], dim=-1)[None, :, :, None, :]  # [1,h,w,1,2]
    kpz = torch.cat([torch.zeros(b, 1, kp, 2, dtype=keypoints_yx.dtype), keypoints_yx], dim=1)  # prepend background
    bi = torch.arange(b)[:, None, None]
    tgt = kpz[bi, labels.to(torch.int64)]  # [b,h,w,kp,2]
    d = (tgt - grid) * (labels != 0)[..., None, None].to(keypoints_yx.dtype)
    nrm = torch.sqrt((d * d).sum(-1, keepdim=True))
    d = d / torch.clamp(nrm, min=1e-12)  # tf.math.l2_normalize
    return d.reshape(b, h, w, kp * 2)


def smooth_l1(e):
    return torch.where(e < 1.0, 0.5 * e * e, e - 0.5)


def losses(output: torch.Tensor, labels: torch.Tensor, keypoints_yx: torch.Tensor, seg_dim: int, kp: int = 9,
           filter_vertex_with_segmentation: bool = True, filter_high_proxy_errors: bool = False):
    """mask / vertex / proxy losses of compute_loss (train_casapose.py:40-145) for the merged-output
    models (not `pvnet`): returns (mask_loss, vertex_loss, proxy_loss).
      mask   : mean softmax cross-entropy (:59-60)
      vertex : smooth_l1_loss(pred, target field, fg weights, normalised per image by ver_dim*sum(w)+1e-3), mean over images
               (loss_functions.py:14-44, ver_dim = 2*kp)
      proxy  : proxy_voting_loss_v2(loss_per_object=False) (:132-203)
    With filter_vertex_with_segmentation the foreground is restricted to pixels whose predicted label
    equals the ground truth (:62-69); the filtered target is a constant (stop_gradient, :95)."""
    b, h, w, _ = output.shape
    logits = output[..., :seg_dim]
    dirs = output[..., seg_dim : seg_dim + 2 * kp]
    lab = labels.to(torch.int64)
    mask_loss = F.cross_entropy(logits.reshape(-1, seg_dim), lab.reshape(-1), reduction="mean")
    fg_lab = lab.clone()
    if filter_vertex_with_segmentation:
        pred = torch.argmax(logits.detach(), dim=-1)
        fg_lab = torch.where(pred == lab, lab, torch.zeros_like(lab))
    if filter_high_proxy_errors:
        # proxy_voting_dist per object (loss_functions.py:47-129) on the filtered mask; objects with value >= 5 leave the foreground
        # (train_casapose.py:71-93); everything here is a constant of the gradient
        with torch.no_grad():
            vd = output[..., seg_dim:seg_dim + 2 * kp].reshape(b, h, w, kp, 2)
            bi0 = torch.arange(b)[:, None, None]
            kk = keypoints_yx[bi0, torch.clamp(fg_lab - 1, min=0)]
            yy0, xx0 = torch.meshgrid(torch.arange(h, dtype=output.dtype) + 0.5, torch.arange(w, dtype=output.dtype) + 0.5, indexing="ij")
            num0 = torch.abs(vd[..., 0] * (kk[..., 1] - xx0[None, :, :, None]) - vd[..., 1] * (kk[..., 0] - yy0[None, :, :, None]))
            nr0 = torch.sqrt((vd * vd).sum(-1))
            dist0 = torch.where(nr0 > 0, num0 / torch.where(nr0 > 0, nr0, torch.ones_like(nr0)), torch.zeros_like(num0))
            per_px = smooth_l1(dist0).sum(-1) * (fg_lab != 0)
            keep = torch.ones(b, seg_dim, dtype=torch.bool)
            for o in range(1, seg_dim):
                m = fg_lab == o
                cnt = m.reshape(b, -1).sum(1).to(output.dtype)
                val = torch.where(cnt >= 20, (per_px * m).reshape(b, -1).sum(1) / (kp * cnt + 1e-3), torch.zeros_like(cnt))
                keep[:, o] = val < 5
            fg_lab = torch.where(keep[bi0, fg_lab], fg_lab, torch.zeros_like(fg_lab))
    wgt = (fg_lab != 0).to(output.dtype)  # [b,h,w]
    target = target_vector_field(labels, keypoints_yx)
    # vertex loss
    e = torch.abs(wgt[..., None] * (dirs - target))
    per_img = smooth_l1(e).reshape(b, -1).sum(1) / (2 * kp * wgt.reshape(b, -1).sum(1) + 1e-3)
    vertex_loss = per_img.mean()
    # proxy voting loss: distance between the GT keypoint and the line through the pixel along the predicted direction
    v = dirs.reshape(b, h, w, kp, 2)
    vy, vx = v[..., 0], v[..., 1]
    bi = torch.arange(b)[:, None, None]
    obj = torch.clamp(fg_lab - 1, min=0)  # argmax over target_seg[...,1:] (background pixels have weight 0 anyway)
    k = keypoints_yx[bi, obj]  # [b,h,w,kp,2] (y,x)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=output.dtype) + 0.5, torch.arange(w, dtype=output.dtype) + 0.5, indexing="ij")
    num = torch.abs(vy * (k[..., 1] - xx[None, :, :, None]) - vx * (k[..., 0] - yy[None, :, :, None]))
    nrm = torch.sqrt(vy * vy + vx * vx)
    dist = torch.where(nrm > 0, num / torch.where(nrm > 0, nrm, torch.ones_like(nrm)), torch.zeros_like(num))  # divide_no_nan
    dist = torch.abs(wgt[..., None] * dist)
    per_img = smooth_l1(dist).reshape(b, -1).sum(1) / (2 * kp * wgt.reshape(b, -1).sum(1) + 1e-3)
    proxy_loss = per_img.mean()
    return mask_loss, vertex_loss, proxy_loss


# --------------------------------------------------------------------------------------
#  keypoint reprojection loss (utils/loss_functions.py:207-344) through the LS voter
# --------------------------------------------------------------------------------------
def ls_voting(labels: torch.Tensor, dirs: torch.Tensor, conf: torch.Tensor, objects: int) -> torch.Tensor:
    """CoordLSVotingWeighted.calc (voting_layers_2d.py:83-122) with a hard label map, differentiable in
    dirs/conf.  labels [B,H,W] int, dirs [B,H,W,2*kp] (dy,dx), conf [B,H,W,kp] -> [B,objects,kp,2] (y,x) px."""
    b, h, w, kp = conf.shape
    dt = dirs.dtype
    wgt = F.softplus(conf)                                            # :35
    d = dirs.reshape(b, h, w, kp, 2)
    s2 = (d * d).sum(-1, keepdim=True)
    pos = s2 > 0
    nrm = torch.sqrt(torch.where(pos, s2, torch.ones_like(s2)))        # keeps autograd finite at d = 0
    n = torch.where(pos, d / nrm, torch.zeros_like(d))                 # divide_no_nan :90
    eye = torch.eye(2, dtype=dt)
    R = (eye - n[..., :, None] * n[..., None, :]) * wgt[..., None, None]   # [b,h,w,kp,2,2]
    yy, xx = torch.meshgrid((torch.arange(h, dtype=dt) + 0.5) / h, (torch.arange(w, dtype=dt) + 0.5) / h, indexing="ij")
    c = torch.stack([yy, xx], dim=-1)[None, :, :, None, :, None]       # (y,x)/H :99-101
    q = (R @ c)[..., 0]                                                # [b,h,w,kp,2]
    out = []
    for o in range(objects):
        m = (labels == o + 1).to(dt)[..., None, None]
        A = (R * m[..., None]).sum(dim=(1, 2))                         # [b,kp,2,2]
        t = (q * m).sum(dim=(1, 2))                                    # [b,kp,2]
        # tf.linalg.pinv's default cut-off 10 * max(rows, cols) * eps (voting_layers_2d.py:116), not torch's 1e-15
        out.append((torch.linalg.pinv(A, rtol=10.0 * 2 * torch.finfo(dt).eps) @ t[..., None])[..., 0] * h)
    return torch.stack(out, dim=1)


def ransac_round_counts(direct: torch.Tensor, coords: torch.Tensor, idxs: torch.Tensor, thresh: float = 0.99) -> torch.Tensor:
    """One RANSAC round of ransac_voting_batch for one object (ransac_voting.py:197-249,319-329): hypotheses from the drawn pixel
    pairs, then the [hn,tn,vn] cosine tests and their inlier counts [hn,vn].  direct [tn,vn,2] (dx,dy), coords [tn,2] (x,y),
    idxs [hn,vn,2] int64.  The broadcast form of the reference, on torch-CPU threads: the CPU-baseline leg of `bench.py --mode vote`."""
    hn, vn, _ = idxs.shape
    vi = torch.arange(vn)[None, :, None]
    c_s, d_s = coords[idxs], direct[idxs, vi]                                                     # [hn,vn,2,2]
    det = d_s[:, :, 1, 0] * d_s[:, :, 0, 1] - d_s[:, :, 1, 1] * d_s[:, :, 0, 0]
    u = ((c_s[:, :, 1, 1] - c_s[:, :, 0, 1]) * d_s[:, :, 1, 0] - (c_s[:, :, 1, 0] - c_s[:, :, 0, 0]) * d_s[:, :, 1, 1]) / det
    hyp = torch.where((det.abs() > 1e-6)[..., None], c_s[:, :, 0] + d_s[:, :, 0] * u[..., None], torch.zeros_like(c_s[:, :, 0]))
    hd = hyp[:, None] - coords[None, :, None]                                                     # [hn,tn,vn,2]
    nd = torch.sqrt((direct * direct).sum(-1))[None]
    nh = torch.sqrt((hd * hd).sum(-1))
    valid = (nd > 1e-6) & (nh > 1e-6) & (hyp.sum(-1).abs() > 1e-6)[:, None, :]
    ang = (direct[None] * hd).sum(-1) / (nd * nh)
    return (valid & (ang > thresh)).sum(dim=1)


def crop_to_image_affine(offsets: np.ndarray) -> np.ndarray:
    """2x3 matrices of transform_points_back_tf_batch (ransac_voting.py:124-158) acting on crop pixels (x,y);
    offsets [B,10] = [h_crop, w_crop, ?, ?, dx, dy, angle_deg, scale, sx, sy] (loss_functions.py:276-285)."""
    o = np.asarray(offsets, np.float64)
    hc, wc, dx, dy, ang, sc, sx, sy = o[:, 0], o[:, 1], o[:, 4], o[:, 5], o[:, 6], o[:, 7], o[:, 8], o[:, 9]
    ar = -ang * (np.pi / 180.0)
    a, b = np.cos(ar), np.sin(ar)
    cx, cy = sx / 2.0, sy / 2.0
    c = (1.0 - a) * cx - b * cy
    d = b * cx + (1.0 - a) * cy
    A = np.zeros((o.shape[0], 2, 3))
    A[:, 0, 0], A[:, 0, 1], A[:, 0, 2] = a / sc, b / sc, a * (wc - dx) + b * (hc - dy) + c
    A[:, 1, 0], A[:, 1, 1], A[:, 1, 2] = -b / sc, a / sc, -b * (wc - dx) + a * (hc - dy) + d
    return A


def project_points(xyz: np.ndarray, K: np.ndarray, RT: np.ndarray) -> np.ndarray:
    """project_tf_batch (ransac_voting.py:185-194): xyz [N,kp,3], K [3,3], RT [N,3,4] -> [N,kp,2] (x,y)."""
    cam = xyz @ np.transpose(RT[:, :, :3], (0, 2, 1)) + np.transpose(RT[:, :, 3:], (0, 2, 1))
    pix = cam @ K.T
    z = pix[..., 2:]
    return np.where(z != 0, pix[..., :2] / np.where(z != 0, z, 1.0), 0.0)


def keypoint_reprojection_loss(coords_yx: torch.Tensor, gt_xy: torch.Tensor, affine: torch.Tensor, avail: torch.Tensor,
                               conf: torch.Tensor, labels_gt: torch.Tensor, max_pixel_error: float = 25.0,
                               confidence_regularization: bool = False) -> torch.Tensor:
    """keypoint_reprojection_loss without BPnP (loss_functions.py:207-344).  coords_yx [B,oc,kp,2] from the voter,
    gt_xy [B,oc,kp,2] projected GT keypoints (image px), affine [B,2,3], avail [B,oc] in {0,1}."""
    b, oc, kp, _ = coords_yx.shape
    xy = coords_yx.flip(-1)                                             # tf.reverse :229
    X = affine[:, None, None, 0, 0] * xy[..., 0] + affine[:, None, None, 0, 1] * xy[..., 1] + affine[:, None, None, 0, 2]
    Y = affine[:, None, None, 1, 0] * xy[..., 0] + affine[:, None, None, 1, 1] * xy[..., 1] + affine[:, None, None, 1, 2]
    pts = torch.stack([X, Y], dim=-1) * avail[..., None, None]
    gt = gt_xy * avail[..., None, None]
    s2 = ((gt - pts) ** 2).sum(-1)
    e = torch.where(s2 > 0, torch.sqrt(torch.where(s2 > 0, s2, torch.ones_like(s2))), torch.zeros_like(s2))
    l = torch.where(e < 1.0, 0.5 * e * e, e - 0.5)
    l = torch.where(l > max_pixel_error, max_pixel_error + (l - max_pixel_error) * 0.01, l)
    l = (l * avail[..., None]).mean(dim=2)                              # mean over keypoints
    na = avail.sum()
    loss = torch.where(na > 0, l.sum() / torch.where(na > 0, na, torch.ones_like(na)), torch.zeros_like(na))
    if confidence_regularization:
        fg = (labels_gt != 0).to(conf.dtype)[..., None]
        cs = (F.softplus(conf) * fg).sum(dim=(1, 2))                    # [b,kp]
        ms = fg.sum(dim=(1, 2))                                         # [b,1]
        cl = torch.where(ms > 0, cs / torch.where(ms > 0, ms, torch.ones_like(ms)), torch.zeros_like(cs))
        loss = loss + torch.abs(cl - 0.7).mean()
    return loss


# --------------------------------------------------------------------------------------
#  inference-only fast path of the SAME graph (bench.py's CPU baseline, round 4)
# --------------------------------------------------------------------------------------
def prepare_inference(p: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Constant folding for forward_infer_fast: every (Sync)BatchNormalization becomes one per-channel affine (scale, shift); every CLADE layer a
    per-class table of them; convolution kernels go to OIHW channels-last once; the partial convolutions' [Cin,3,3,Cout] weights to ONE
    1x1 kernel with 9*Cout outputs (tap-major).  Values only -- the arithmetic per output element is the reference's, re-associated."""
    q: Dict[str, torch.Tensor] = {}
    for k, v in p.items():
        if k.endswith(".moving_variance"):
            n = k[: -len(".moving_variance")]
            rstd = 1.0 / torch.sqrt(v + BN_EPS)
            g, b = p.get(n + ".gamma"), p.get(n + ".beta")
            scale = rstd if g is None else g * rstd                   # CLADE: gamma [classes, C] -> table [classes, C]
            shift = -p[n + ".moving_mean"] * scale
            q[n + ".scale"], q[n + ".shift"] = scale.contiguous(), (shift if b is None else shift + b).contiguous()
        elif k.endswith(".kernel"):
            q[k] = v.permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last)
        elif k.endswith(".weights"):                                  # IHWO -> [9*Cout, Cin, 1, 1], output channel = tap * Cout + o
            cin, _, _, cout = v.shape
            q[k] = v.permute(1, 2, 3, 0).reshape(9 * cout, cin, 1, 1).contiguous(memory_format=torch.channels_last)
    return q


def forward_infer_fast(q: Dict[str, torch.Tensor], img: torch.Tensor, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """casapose_c_gcu5 inference with the estimated mask (forward_train(..., labels=None, training=False)) as a CPU program someone would
    actually run: NCHW tensors in channels-last memory end to end (oneDNN's native layout, no per-layer permutes), folded normalisation
    (one fused multiply-add per layer), in-place activations, and the partial convolution as ONE 1x1 convolution to 9*Cout tap planes
    followed by nine masked shifted accumulations on the OUTPUT side (the tap mask is a per-pixel scalar, so (m x) W = m (x W), and every
    decoder layer has Cout <= Cin).  img [B,H,W,3] -> [B,H,W,K+ver_dim]."""  # (test infrastructure: tests/, bench.cpu_baseline, smoke() only)
    def affine(name, x, act):   # act: 0 none, 1 relu, 2 leaky pair relu(t) - relu(-0.1 t) = leaky_relu(t, 0.1)
        y = torch.addcmul(q[name + ".shift"].view(1, -1, 1, 1), x, q[name + ".scale"].view(1, -1, 1, 1))
        return F.relu_(y) if act == 1 else (F.leaky_relu_(y, 0.1) if act == 2 else y)

    def conv(name, x, stride=1, dil=1, pad=0):
        return F.conv2d(x, q[name + ".kernel"], stride=stride, dilation=dil, padding=pad)

    x = img.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    img_c = x
    x = conv("conv0", affine("bn_data", x, 0), stride=2, pad=3)
    x2s = affine("bn0", x, 1)
    x = F.max_pool2d(F.pad(x2s, (1, 1, 1, 1)), 3, 2)
    taps = []
    for s in range(4):
        d = STAGE_DILATION[s]
        for u in range(2):
            base = "stage%d_unit%d_" % (s + 1, u + 1)
            stride = STAGE_STRIDE[s] if u == 0 else 1
            a = affine(base + "bn1", x, 1)
            shortcut = conv(base + "sc", a, stride=stride) if u == 0 else x
            y = affine(base + "bn2", conv(base + "conv1", a, stride=stride, dil=d, pad=d), 1)
            x = conv(base + "conv2", y, dil=d, pad=d).add_(shortcut)
            if u == 0 and s > 0:
                taps.append(a)
    x32s = affine("bn1", x, 1)
    x4s, x8s, _ = taps
    skips = [None, x8s, x4s, x2s, img_c]
    d1 = None
    for i in range(5):
        n = "pv_block_%d" % (i + 1)
        inp = x32s if i == 0 else torch.cat([d1, skips[i]], dim=1)
        y = affine(n + "_bn", conv(n + "_conv2d", inp, pad=1), 1 if i == 0 else 2)
        d1 = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False) if 0 < i < 4 else y
    logits = conv("pv_final_conv_segmentation", d1)
    # `labels` [B,H,W] (optional): condition decoder 2 on a GIVEN hard label map instead of the arg-max of these logits -- how the bench compares
    # a device's vector field with fp64 without the arg-max ties of random weights deciding the comparison
    labs = labels_pyramid(torch.argmax(logits, dim=1) if labels is None else labels.to(torch.int64))   # [B,H,W] int64 and its three [::2, ::2] levels
    lvl = [3, 3, 2, 1, 0]

    def partial(name, x, lab):
        b, _, h, w = x.shape
        zp = F.conv2d(F.pad(x, (1, 1, 1, 1)), q[name + ".weights"])  # [B, 9*Cout, H+2, W+2]: the zero ring of x gives the zero ring of the tap planes
        cout = zp.shape[1] // 9
        lp = F.pad(lab + 1, (1, 1, 1, 1))                            # 0 = outside the image
        out = torch.zeros(b, cout, h, w, dtype=x.dtype).contiguous(memory_format=torch.channels_last)
        cnt = torch.zeros(b, 1, h, w, dtype=x.dtype)
        for t in range(9):
            ky, kx = divmod(t, 3)
            m = (lp[:, ky:ky + h, kx:kx + w] == (lab + 1)).to(x.dtype).unsqueeze(1)
            cnt += m
            out.addcmul_(zp[:, t * cout:(t + 1) * cout, ky:ky + h, kx:kx + w], m)
        return out.mul_(9.0 / cnt)

    d2 = None
    for i in range(5):
        n = "pv_block_%d" % (i + 6)
        lab = labs[lvl[i]]
        inp = x32s if i == 0 else torch.cat([d2, skips[i]], dim=1)
        y = partial(n + "_prepare_conv2d", inp, lab)
        sc = q[n + "_clade.scale"][lab].permute(0, 3, 1, 2)          # per-label rows of the folded CLADE table
        sh = q[n + "_clade.shift"][lab].permute(0, 3, 1, 2)
        y = torch.addcmul(sh, y, sc)
        y = F.relu_(y) if i == 0 else F.leaky_relu_(y, 0.1)
        if 0 < i < 4:
            hi = labs[lvl[i] - 1]
            if i in (1, 2, 3):                                        # guided upsampling (casapose_c_gcu5: blocks 7-9)
                sel = guided_select(lab, hi)
                b, c, h2, w2 = y.shape
                yy, xx = torch.meshgrid(torch.arange(2 * h2), torch.arange(2 * w2), indexing="ij")
                sy = torch.clamp(yy[None] // 2 + sel // 2, max=h2 - 1)
                sx = torch.clamp(xx[None] // 2 + sel % 2, max=w2 - 1)
                bi = torch.arange(b)[:, None, None]
                y = y.permute(0, 2, 3, 1)[bi, sy, sx, :].permute(0, 3, 1, 2)
        d2 = y
    vertex = conv("pv_final_conv_vertex", d2)
    return torch.cat([logits, vertex], dim=1).permute(0, 2, 3, 1)


def ls_voting_fast(labels: torch.Tensor, dirs: torch.Tensor, conf: torch.Tensor, objects: int) -> torch.Tensor:
    """ls_voting (CoordLSVotingWeighted.calc, voting_layers_2d.py:83-122) as ONE pass: the five distinct entries of w (I - n n^T) and of its
    product with the pixel centre are formed per pixel and keypoint in the input precision, accumulated per OBJECT in fp64 with index_add
    (the reference's fp64 reduce_sum over the masked map, :110-111), then the 2x2 systems are solved with TensorFlow's pinv cut-off.
    labels [B,H,W] int (0 = background), dirs [B,H,W,2*kp] (dy,dx), conf [B,H,W,kp] -> [B,objects,kp,2] (y,x) pixels.  No autograd."""
    b, h, w, kp = conf.shape
    dt = dirs.dtype
    wgt = F.softplus(conf)
    d = dirs.reshape(b, h, w, kp, 2)
    nrm = torch.sqrt((d * d).sum(-1, keepdim=True))
    n = torch.where(nrm > 0, d / torch.where(nrm > 0, nrm, torch.ones_like(nrm)), torch.zeros_like(d))
    ny, nx = n[..., 0], n[..., 1]
    r00, r01, r11 = (1.0 - ny * ny) * wgt, (-ny * nx) * wgt, (1.0 - nx * nx) * wgt
    cy = ((torch.arange(h, dtype=dt) + 0.5) / h).view(1, h, 1, 1)
    cx = ((torch.arange(w, dtype=dt) + 0.5) / h).view(1, 1, w, 1)
    terms = torch.stack([r00, r01, r11, r00 * cy + r01 * cx, r01 * cy + r11 * cx], dim=-1).reshape(b, h * w, kp * 5).double()
    idx = labels.reshape(b, h * w).to(torch.int64)
    acc = torch.zeros(b, objects + 1, kp * 5, dtype=torch.float64)
    for i in range(b):
        acc[i].index_add_(0, idx[i], terms[i])
    s = acc[:, 1:].reshape(b, objects, kp, 5)
    A = torch.stack([torch.stack([s[..., 0], s[..., 1]], -1), torch.stack([s[..., 1], s[..., 2]], -1)], -2)
    t = s[..., 3:5]
    sol = (torch.linalg.pinv(A, rtol=10.0 * 2 * torch.finfo(torch.float64).eps) @ t[..., None])[..., 0] * h
    return sol.to(dt)
