"""CPU oracle for the CASAPose hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy restatement (fp64 by default) of the arithmetic the reference
performs on the path  image -> ResNet-18(OS8) -> two decoders -> keypoint voting.
It exists so the HIP kernels in ``casapose_amd/csrc`` have something independent to
be compared with.  It is NOT part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it, and only as the checker.

PARITY UNPINNED: the reference is pure Python on TensorFlow 2.9.1 / tensorflow-addons
0.17.0 / OpenCV 4.5.5 (requirements.txt:1-3), none of which can be imported in the
build container, and the reference ships no tests, golden vectors or weights
(SURVEY.md F2/F3).  The functions below therefore restate the *published semantics*
of the TF ops at the reference's call sites; the hand-derived known-answer tests in
``tests/test_oracle_kat.py`` pin those semantics, and ``SURVEY.md`` Appendix B lists
the assumptions (B1-B12) that must be re-verified the day a TF box is available.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Layout is NHWC everywhere, like the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

BN_EPS = 2e-5  # casapose/pose_models/models/resnet.py:44, _normalization_layers.py:108

# --------------------------------------------------------------------------------------
#  primitive layers
# --------------------------------------------------------------------------------------


def zero_pad(x: np.ndarray, p: int) -> np.ndarray:
    """layers.ZeroPadding2D(padding=(p, p)) -- resnet.py:96,102,248,253; casapose.py:71."""
    if p == 0:
        return x
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))


def conv2d_valid(x: np.ndarray, w_hwio: np.ndarray, stride: int = 1, dilation: int = 1) -> np.ndarray:
    """layers.Conv2D(..., padding="valid", use_bias=False) on NHWC with HWIO kernel
    (resnet.py:29-36,85-87,97-99,103,249).  Cross-correlation (no kernel flip)."""
    n, h, w, c = x.shape
    kh, kw, ci, co = w_hwio.shape
    assert ci == c, (ci, c)
    eh = (kh - 1) * dilation + 1
    ew = (kw - 1) * dilation + 1
    ho = (h - eh) // stride + 1
    wo = (w - ew) // stride + 1
    out = np.zeros((n, ho, wo, co), dtype=np.result_type(x, w_hwio))
    for ky in range(kh):
        for kx in range(kw):
            ys = ky * dilation
            xs = kx * dilation
            patch = x[:, ys : ys + (ho - 1) * stride + 1 : stride, xs : xs + (wo - 1) * stride + 1 : stride, :]
            out += patch.reshape(-1, c).dot(w_hwio[ky, kx]).reshape(n, ho, wo, co)
    return out


def conv2d(x, w_hwio, stride=1, dilation=1, pad=0):
    return conv2d_valid(zero_pad(x, pad), w_hwio, stride, dilation)


def batchnorm_inference(x, gamma, beta, mean, var, eps=BN_EPS):
    """(Sync)BatchNormalization with training=False: y = gamma*(x-mean)/sqrt(var+eps)+beta
    (resnet.py:39-49,78,100,247,250,303; casapose.py:77).  gamma/beta may be None
    (scale=False / center=False)."""
    y = (x - mean) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return y


def relu(x):
    return np.maximum(x, 0)


def leaky_as_relu_pair(x):
    """casapose.py:98-105: relu(x) - relu(-0.1*x)."""
    return np.maximum(x, 0) - np.maximum(-0.1 * x, 0)


def maxpool_3x3_s2_pad1(x):
    """ZeroPadding2D(1) then MaxPooling2D((3,3), strides 2, 'valid') -- resnet.py:253-254.
    The padding value is ZERO (not -inf)."""
    xp = zero_pad(x, 1)
    n, h, w, c = xp.shape
    ho = (h - 3) // 2 + 1
    wo = (w - 3) // 2 + 1
    out = np.full((n, ho, wo, c), -np.inf, dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, xp[:, ky : ky + (ho - 1) * 2 + 1 : 2, kx : kx + (wo - 1) * 2 + 1 : 2, :])
    return out


def upsample_bilinear_x2(x):
    """tf.compat.v1.keras.layers.UpSampling2D(size 2, interpolation="bilinear")
    (casapose.py:135-140) == tf.image.resize(bilinear, half-pixel centres, no antialias)
    -- SURVEY Appendix B4.  out(Y) samples source coordinate (Y+0.5)/2-0.5 with edge clamp."""
    n, h, w, c = x.shape

    def taps(size):
        dst = np.arange(2 * size)
        src = (dst + 0.5) / 2.0 - 0.5
        i0 = np.floor(src).astype(np.int64)
        frac = src - i0
        lo = np.clip(i0, 0, size - 1)
        hi = np.clip(i0 + 1, 0, size - 1)
        return lo, hi, frac

    ylo, yhi, fy = taps(h)
    xlo, xhi, fx = taps(w)
    fy = fy[None, :, None, None]
    fx = fx[None, None, :, None]
    top = x[:, ylo][:, :, xlo] * (1 - fx) + x[:, ylo][:, :, xhi] * fx
    bot = x[:, yhi][:, :, xlo] * (1 - fx) + x[:, yhi][:, :, xhi] * fx
    return top * (1 - fy) + bot * fy


def upsample_nearest_x2(x):
    return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def half_size(mask):
    """HalfSize: identity-initialised 1x1 stride-2 VALID conv == mask[:, ::2, ::2, :]
    (_normalization_layers.py:258-299); output size int(0.5*h) x int(0.5*w)."""
    n, h, w, k = mask.shape
    return mask[:, : (h // 2) * 2 : 2, : (w // 2) * 2 : 2, :]


def saturated_softmax(logits):
    """softmax(1e6 * logits) -- pose_models.py:547-549; voting_layers_2d.py:38-39.
    Computed stably; exactly one-hot unless the top logits are within ~1e-5."""
    z = logits.astype(np.float64) * 1e6
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=-1, keepdims=True)).astype(logits.dtype)


def onehot_from_labels(labels: np.ndarray, num_classes: int, dtype=np.float64) -> np.ndarray:
    return (labels[..., None] == np.arange(num_classes)).astype(dtype)


# --------------------------------------------------------------------------------------
#  class-adaptive decoder layers
# --------------------------------------------------------------------------------------


def _extract_3x3_same(m):
    """[b,h,w,k] -> [b,h,w,9,k], tap n = ky*3+kx, zero outside (SAME) -- the layout
    assumed for tf.image.extract_patches (_normalization_layers.py:334-341)."""
    b, h, w, k = m.shape
    mp = zero_pad(m, 1)
    out = np.empty((b, h, w, 9, k), dtype=m.dtype)
    for ky in range(3):
        for kx in range(3):
            out[:, :, :, ky * 3 + kx, :] = mp[:, ky : ky + h, kx : kx + w, :]
    return out


def partial_conv_mask(seg_mask):
    """m(p,n) and norm(p) of PartialConvolution.calc (_normalization_layers.py:333-352).

    m(p,n) = sum_k [seg_k(p) == max_k seg(p)] * seg_k(p+n);  norm = 9 / count_nonzero_n m(p,n)
    (divide_no_nan)."""
    s_patch = _extract_3x3_same(seg_mask)  # [b,h,w,9,k]
    s_max = seg_mask.max(axis=-1, keepdims=True)[:, :, :, None, :]
    sel = np.where(seg_mask[:, :, :, None, :] == s_max, s_patch, 0.0)
    m = sel.sum(axis=-1)  # [b,h,w,9]
    cnt = np.count_nonzero(m, axis=-1).astype(seg_mask.dtype)[..., None]
    norm = np.where(cnt > 0, 9.0 / np.where(cnt > 0, cnt, 1.0), 0.0)
    return m, norm


def partial_convolution(x, w_chwo, seg_mask=None):
    """PartialConvolution (_normalization_layers.py:302-377).  Weight is [Cin,3,3,Cout].
    One input -> ordinary SAME 3x3 conv (:327-331); with a mask ->
    out(p,o) = norm(p) * sum_{c,n} x(p+n,c) * m(p,n) * W[c,n,o]  (:364-371)."""
    ci, kh, kw, co = w_chwo.shape
    assert kh == 3 and kw == 3
    w_hwio = np.transpose(w_chwo, (1, 2, 0, 3))
    if seg_mask is None:
        return conv2d(x, w_hwio, pad=1)
    b, h, w, c = x.shape
    m, norm = partial_conv_mask(seg_mask)
    xp = zero_pad(x, 1)
    out = np.zeros((b, h, w, co), dtype=np.result_type(x, w_chwo))
    for ky in range(3):
        for kx in range(3):
            tap = xp[:, ky : ky + h, kx : kx + w, :] * m[:, :, :, ky * 3 + kx, None]
            out += tap.reshape(-1, c).dot(w_hwio[ky, kx]).reshape(b, h, w, co)
    return out * norm


def clade_weighted(x, seg_mask, gamma_kc, beta_kc, mean, var, eps=BN_EPS):
    """ClassAdaptiveWeightedNormalization.calc (_normalization_layers.py:119-139):
    gamma1 = seg . gamma, beta1 = seg . beta (tensordot over the class axis);
    y = gamma1 * BN_noaffine(x) + beta1."""
    g1 = np.tensordot(seg_mask, gamma_kc, axes=([3], [0]))
    b1 = np.tensordot(seg_mask, beta_kc, axes=([3], [0]))
    xn = batchnorm_inference(x, None, None, mean, var, eps)
    return g1 * xn + b1


def _float_label(seg):
    """label(p) = sum_k [seg_k == max] * seg_k * (k+1) (_normalization_layers.py:512-531)."""
    k = seg.shape[-1]
    r_up = np.arange(1, k + 1, dtype=seg.dtype)
    mx = seg.max(axis=-1, keepdims=True)
    return (np.where(seg == mx, seg, 0.0) * r_up).sum(axis=-1)


def _patch_2x2_same(lab):
    """2x2 'SAME' patches: TF pads bottom/right only ->
    neighbours {(0,0),(0,1),(1,0),(1,1)} (_normalization_layers.py:534-539)."""
    b, h, w = lab.shape
    lp = np.pad(lab, ((0, 0), (0, 1), (0, 1)))
    return np.stack([lp[:, :h, :w], lp[:, :h, 1 : w + 1], lp[:, 1 : h + 1, :w], lp[:, 1 : h + 1, 1 : w + 1]], axis=-1)


def guided_upsampling_select(seg_d, seg_u):
    """Index of the low-res neighbour chosen for every hi-res pixel by GuidedUpsampling
    (_normalization_layers.py:507-557): the FIRST of {(y,x),(y,x+1),(y+1,x),(y+1,x+1)}
    whose low-res label equals the hi-res label, else (y,x).  Returns int [b,2h,2w] in 0..3."""
    lab_d = _float_label(seg_d)
    lab_u = _float_label(seg_u)
    b, h2, w2 = lab_d.shape
    patches = _patch_2x2_same(lab_d)  # [b,h2,w2,4]
    pu = np.repeat(np.repeat(patches, 2, axis=1), 2, axis=2)  # [b,2h2,2w2,4]
    eq = pu == lab_u[:, : 2 * h2, : 2 * w2, None]
    r_down = np.array([4.0, 3.0, 2.0, 1.0])
    score = eq * r_down
    sel = np.argmax(score, axis=-1)  # all-zero -> 0 == first
    return sel


def guided_upsampling(x, seg_d, seg_u):
    """GuidedUpsampling.call (_normalization_layers.py:507-566)."""
    b, h2, w2, c = x.shape
    sel = guided_upsampling_select(seg_d, seg_u)
    yy, xx = np.meshgrid(np.arange(2 * h2), np.arange(2 * w2), indexing="ij")
    sy = yy[None] // 2 + sel // 2
    sx = xx[None] // 2 + sel % 2
    # a selected neighbour is never a padded one (padded label 0 never matches a label >= 1)
    sy = np.minimum(sy, h2 - 1)
    sx = np.minimum(sx, w2 - 1)
    bi = np.arange(b)[:, None, None]
    return x[bi, sy, sx, :]


def guided_bilinear_upsampling(x, seg_d, seg_u):
    """GuidedBilinearUpsampling.call (_normalization_layers.py:607-664): taps with a foreign
    label are replaced by the mean of the matching taps; fixed per-sub-pixel weights."""
    lab_d = _float_label(seg_d)
    lab_u = _float_label(seg_u)
    b, h2, w2, c = x.shape
    patches = _patch_2x2_same(lab_d)
    pu = np.repeat(np.repeat(patches, 2, axis=1), 2, axis=2)
    cond = pu == lab_u[:, : 2 * h2, : 2 * w2, None]  # [b,H,W,4]
    xp = np.pad(x, ((0, 0), (0, 1), (0, 1), (0, 0)))
    xt = np.stack(
        [xp[:, :h2, :w2], xp[:, :h2, 1 : w2 + 1], xp[:, 1 : h2 + 1, :w2], xp[:, 1 : h2 + 1, 1 : w2 + 1]], axis=3
    )  # [b,h2,w2,4,c]
    xt = np.repeat(np.repeat(xt, 2, axis=1), 2, axis=2)  # [b,H,W,4,c]
    cf = cond[..., None]
    xm = np.where(cf, xt, 0.0)
    norm = cond.sum(axis=-1, keepdims=True)[..., None].astype(x.dtype)
    mean = np.where(norm > 0, xm.sum(axis=3, keepdims=True) / np.where(norm > 0, norm, 1.0), 0.0)
    xf = np.where(cf, xm, mean)
    interp = np.array(
        [[1.0, 0.0, 0.0, 0.0], [0.5, 0.5, 0.0, 0.0], [0.5, 0.0, 0.5, 0.0], [0.25, 0.25, 0.25, 0.25]], dtype=x.dtype
    )
    yy, xx = np.meshgrid(np.arange(2 * h2), np.arange(2 * w2), indexing="ij")
    sub = (yy % 2) * 2 + (xx % 2)
    wts = interp[sub]  # [H,W,4]
    return (xf * wts[None, :, :, :, None]).sum(axis=3)


# --------------------------------------------------------------------------------------
#  parameters: names follow the Keras layer names
# --------------------------------------------------------------------------------------

STAGE_FILTERS = (64, 128, 256, 512)
STAGE_STRIDE = (1, 2, 1, 1)  # resnet.py:262-290 with output_stride 8
STAGE_DILATION = (1, 1, 2, 4)


def encoder_conv_specs() -> List[Tuple[str, int, int, int]]:
    """(name, k, cin, cout) for the 21 encoder convs of ResNet-18 (resnet.py:246-305)."""
    specs = [("conv0", 7, 3, 64)]
    cin = 64
    for s, f in enumerate(STAGE_FILTERS):
        for u in range(2):
            base = "stage%d_unit%d_" % (s + 1, u + 1)
            if u == 0:
                specs.append((base + "sc", 1, cin, f))
            specs.append((base + "conv1", 3, cin, f))
            specs.append((base + "conv2", 3, f, f))
            cin = f
    return specs


def encoder_bn_specs() -> List[Tuple[str, int, bool]]:
    """(name, channels, has_gamma)."""
    specs = [("bn_data", 3, False), ("bn0", 64, True)]
    cin = 64
    for s, f in enumerate(STAGE_FILTERS):
        for u in range(2):
            base = "stage%d_unit%d_" % (s + 1, u + 1)
            specs.append((base + "bn1", cin, True))
            specs.append((base + "bn2", f, True))
            cin = f
    specs.append(("bn1", 512, True))
    return specs


DECODER_DIMS = (256, 128, 64, 32, 32)  # fcdim, s8dim, s4dim, s2dim, raw_dim (pose_models.py:516-520)
DECODER_IN = (512, 256 + 128, 128 + 64, 64 + 64, 32 + 3)


VARIANTS = {  # per decoder-2 block 6..10: (partial convolution, guided upsampling); pose_models.py:14-635
    "casapose_c": ((False,) * 5, (False,) * 5),                                    # CASAPoseConditional1: CLADE only, nearest x2
    "casapose_c_gu": ((False,) * 5, (False, True, True, True, False)),            # CASAPoseConditional2
    "casapose_c_gcu3": ((True, True, True, False, False), (False, True, True, True, False)),
    "casapose_c_gcu4": ((True, True, True, True, False), (False, True, True, True, False)),
    "casapose_c_gcu5": ((True,) * 5, (False, True, True, True, False)),
    "casapose_c_gcu4_bilat": ((True, True, True, True, False), (False, True, True, True, False)),   # CASAPoseConditional9
}
BILINEAR_GUIDED = {"casapose_c_gcu4_bilat": (False, True, True, True, False)}  # blocks that use GuidedBilinearUpsampling
# "Alternative models" that share convolution weights between the decoders (pose_models.py:699-1362).  shared[i]: block i+1 of
# decoder 1 is the one-input PartialConvolution `pv_block_{i+1}_{i+6}_conv2d` ([Cin,3,3,Cout] weights, SAME conv) and block i+6
# of decoder 2 uses the same weights with the mask; reuse_first: block 6 has NO convolution of its own -- it normalises the raw
# output y of block 1's convolution (:731,:762-770); skips2 False: decoder 2 takes no skip connections (CASAPoseConditional8).
SHARED = {
    "casapose_c_gcu5_sw5": dict(shared=(True,) * 5, reuse_first=True, skips2=True),                           # CASAPoseConditional6
    "casapose_c_gcu4_sw1": dict(shared=(True, False, False, False, False), reuse_first=True, skips2=True),   # CASAPoseConditional7
    "casapose_c_gcu5_sw1": dict(shared=(True, False, False, False, False), reuse_first=True, skips2=False),  # CASAPoseConditional8
    "casapose_c_gcu4_sw2": dict(shared=(True, True, False, False, False), reuse_first=False, skips2=True),   # CASAPoseConditional10
}
NOT_SHARED = dict(shared=(False,) * 5, reuse_first=False, skips2=True)
VARIANTS.update({
    "casapose_c_gcu5_sw5": ((False, True, True, True, True), (False, True, True, True, False)),
    "casapose_c_gcu4_sw1": ((False, True, True, True, True), (False, True, True, True, False)),
    "casapose_c_gcu5_sw1": ((False, True, True, True, True), (False, True, True, True, False)),
    "casapose_c_gcu4_sw2": ((True, True, True, True, False), (False, True, True, True, False)),
})


def shared_key(i: int) -> str:
    return "pv_block_%d_%d_conv2d.weights" % (i + 1, i + 6)


def init_params(seg_dim: int, ver_dim: int, seed: int = 1237, dtype=np.float32, randomize_norm: bool = True,
                partial=(True,) * 5, pvnet: bool = False, shared=(False,) * 5, reuse_first: bool = False, skips2: bool = True):
    """Random parameters of casapose_c_gcu5 (or, with ``partial``, of a variant whose decoder-2 block i is an ordinary
    convolution `pv_block_N_conv2d.kernel` when partial[i] is False) with the reference's shapes/initialisers
    (he_uniform conv kernels: resnet.py:31; _normalization_layers.py:317).  With
    ``randomize_norm`` the BN/CLADE statistics and affine terms are randomised as in
    SURVEY 8(d) so tests exercise every term."""
    rng = np.random.default_rng(seed)
    p: Dict[str, np.ndarray] = {}

    def he(shape, fan_in):
        lim = math.sqrt(6.0 / fan_in)
        return rng.uniform(-lim, lim, size=shape).astype(dtype)

    def bn(name, c, has_gamma=True, has_beta=True):
        if randomize_norm:
            if has_gamma:
                p[name + ".gamma"] = rng.uniform(0.5, 1.5, c).astype(dtype)
            if has_beta:
                p[name + ".beta"] = (0.1 * rng.standard_normal(c)).astype(dtype)
            p[name + ".moving_mean"] = (0.1 * rng.standard_normal(c)).astype(dtype)
            p[name + ".moving_variance"] = rng.uniform(0.5, 1.5, c).astype(dtype)
        else:
            if has_gamma:
                p[name + ".gamma"] = np.ones(c, dtype)
            if has_beta:
                p[name + ".beta"] = np.zeros(c, dtype)
            p[name + ".moving_mean"] = np.zeros(c, dtype)
            p[name + ".moving_variance"] = np.ones(c, dtype)

    for name, k, ci, co in encoder_conv_specs():
        p[name + ".kernel"] = he((k, k, ci, co), k * k * ci)
    for name, c, has_gamma in encoder_bn_specs():
        bn(name, c, has_gamma=has_gamma)
    for i in range(5):
        ci, co = DECODER_IN[i], DECODER_DIMS[i]
        n1 = "pv_block_%d" % (i + 1)
        if shared[i]:
            p[shared_key(i)] = he((ci, 3, 3, co), 9 * ci)
        else:
            p[n1 + "_conv2d.kernel"] = he((3, 3, ci, co), 9 * ci)
        bn(n1 + "_bn", co)
        n2 = "pv_block_%d" % (i + 6)
        if pvnet:
            continue
        ci2 = ci if (skips2 or i == 0) else DECODER_DIMS[i - 1]
        if shared[i] or (i == 0 and reuse_first):
            pass  # no weights of its own
        elif partial[i]:
            p[n2 + "_prepare_conv2d.weights"] = he((ci2, 3, 3, co), 9 * ci2)
        else:
            p[n2 + "_conv2d.kernel"] = he((3, 3, ci2, co), 9 * ci2)
        bn(n2 + "_clade", co, has_gamma=False, has_beta=False)
        if randomize_norm:
            p[n2 + "_clade.gamma"] = rng.uniform(0.5, 1.5, (seg_dim, co)).astype(dtype)
            p[n2 + "_clade.beta"] = (0.1 * rng.standard_normal((seg_dim, co))).astype(dtype)
        else:
            p[n2 + "_clade.gamma"] = np.ones((seg_dim, co), dtype)
            p[n2 + "_clade.beta"] = np.zeros((seg_dim, co), dtype)
    if pvnet:
        p["pv_final_conv.kernel"] = he((1, 1, 32, seg_dim + ver_dim), 32)
        return p
    p["pv_final_conv_segmentation.kernel"] = he((1, 1, 32, seg_dim), 32)
    p["pv_final_conv_vertex.kernel"] = he((1, 1, 32, ver_dim), 32)
    return p


def _bn(p, name, x):
    return batchnorm_inference(
        x, p.get(name + ".gamma"), p.get(name + ".beta"), p[name + ".moving_mean"], p[name + ".moving_variance"]
    )


# --------------------------------------------------------------------------------------
#  encoder / decoders
# --------------------------------------------------------------------------------------


def residual_unit(p, x, stage, unit):
    """residual_conv_block (resnet.py:57-113).  Returns (out, relu1-activation)."""
    base = "stage%d_unit%d_" % (stage + 1, unit + 1)
    d = STAGE_DILATION[stage]
    stride = STAGE_STRIDE[stage] if unit == 0 else 1
    a = relu(_bn(p, base + "bn1", x))
    if unit == 0:  # cut == "post"
        shortcut = conv2d(a, p[base + "sc.kernel"], stride=stride)
    else:  # cut == "pre"
        shortcut = x
    y = conv2d(a, p[base + "conv1.kernel"], stride=stride, dilation=d, pad=d)
    y = relu(_bn(p, base + "bn2", y))
    y = conv2d(y, p[base + "conv2.kernel"], dilation=d, pad=d)
    return y + shortcut, a


def resnet18_os8(p, img):
    """ResNet(...) body with include_top=False (resnet.py:246-305).
    Returns [x2s, x4s, x8s, x16s, x32s]."""
    x = _bn(p, "bn_data", img)
    x = conv2d(x, p["conv0.kernel"], stride=2, pad=3)
    x2s = relu(_bn(p, "bn0", x))
    x = maxpool_3x3_s2_pad1(x2s)
    taps = []
    for s in range(4):
        for u in range(2):
            x, a = residual_unit(p, x, s, u)
            if u == 0 and s > 0:
                taps.append(a)
    x32s = relu(_bn(p, "bn1", x))
    return [x2s] + taps + [x32s]


def decoder1_block(p, x, idx, leaky, upsample, shared=False, return_raw=False):
    """casa_layer with seg_mask=None (casapose.py:61-77,98-140); `shared`: the convolution is the one-input PartialConvolution
    pv_block_{idx}_{idx+5}_conv2d (an ordinary SAME conv, _normalization_layers.py:327-331) followed by casa_layer(skip_conv=True)."""
    n = "pv_block_%d" % idx
    if shared:
        x = partial_convolution(x, p[shared_key(idx - 1)])
    else:
        x = conv2d(x, p[n + "_conv2d.kernel"], pad=1)
    raw = x
    x = _bn(p, n + "_bn", x)
    x = leaky_as_relu_pair(x) if leaky else relu(x)
    if upsample:
        x = upsample_bilinear_x2(x)
    return (x, raw) if return_raw else x


def decoder2_block(p, x, idx, mask, leaky, guide=None, bilinear_guided=False, partial=True, upsample_nearest=False, shared=False,
                   skip_conv=False):
    """casa_layer with [partial_conv | pad+conv | nothing] + weighted CLADE (+ guided or plain nearest upsampling)
    (casapose.py:61-74,78-82,98-131).  `shared`: the weights are decoder 1's pv_block_{idx-5}_{idx}_conv2d; `skip_conv`: x already is
    the convolution output (casa_layer(skip_conv=True))."""
    n = "pv_block_%d" % idx
    if skip_conv:
        pass
    elif shared:
        x = partial_convolution(x, p[shared_key(idx - 6)], mask if partial else None)
    elif partial:
        x = partial_convolution(x, p[n + "_prepare_conv2d.weights"], mask)
    else:
        x = conv2d(x, p[n + "_conv2d.kernel"], pad=1)
    x = clade_weighted(
        x, mask, p[n + "_clade.gamma"], p[n + "_clade.beta"], p[n + "_clade.moving_mean"], p[n + "_clade.moving_variance"]
    )
    x = leaky_as_relu_pair(x) if leaky else relu(x)
    if guide is not None:
        x = guided_bilinear_upsampling(x, mask, guide) if bilinear_guided else guided_upsampling(x, mask, guide)
    elif upsample_nearest:
        x = upsample_nearest_x2(x)
    return x


def casapose_c_gcu5(p, img, seg_input=None, return_intermediates=False, variant="casapose_c_gcu5"):
    """CASAPoseConditional5 (pose_models.py:513-635) and, with ``variant``, its siblings CASAPoseConditional1-4
    (:14-512) which differ only in which decoder-2 blocks use a partial convolution and guided upsampling.
    img [B,H,W,3]; optional seg_input [B,H,W,K] (the `data_segmentation` input, :550-554).
    Returns [B,H,W,K+ver_dim] = concat(seg logits, vertex)."""
    part, guid = VARIANTS.get(variant, ((True,) * 5, (False,) * 5))
    bil = BILINEAR_GUIDED.get(variant, (False,) * 5)
    sv = SHARED.get(variant, NOT_SHARED)
    sh, reuse_first, skips2 = sv["shared"], sv["reuse_first"], sv["skips2"]
    x2s, x4s, x8s, _x16s, x32s = resnet18_os8(p, img)
    x, y_raw = decoder1_block(p, x32s, 1, leaky=False, upsample=False, shared=sh[0], return_raw=True)
    x = decoder1_block(p, np.concatenate([x, x8s], 3), 2, True, True, shared=sh[1])
    x = decoder1_block(p, np.concatenate([x, x4s], 3), 3, True, True, shared=sh[2])
    x = decoder1_block(p, np.concatenate([x, x2s], 3), 4, True, True, shared=sh[3])
    x = decoder1_block(p, np.concatenate([x, img], 3), 5, True, False, shared=sh[4])
    if variant == "pvnet_combined":  # PVNet (pose_models.py:645-696): one head, no second decoder
        return conv2d(x, p["pv_final_conv.kernel"])
    logits = conv2d(x, p["pv_final_conv_segmentation.kernel"])
    mask = saturated_softmax(logits if seg_input is None else seg_input)
    mask2 = half_size(mask)
    mask4 = half_size(mask2)
    mask8 = half_size(mask4)
    cat = (lambda a, b: np.concatenate([a, b], 3)) if skips2 else (lambda a, b: a)
    if reuse_first:
        y = decoder2_block(p, y_raw, 6, mask8, leaky=False, skip_conv=True)
    else:
        y = decoder2_block(p, x32s, 6, mask8, leaky=False, partial=part[0], shared=sh[0])
    y = decoder2_block(p, cat(y, x8s), 7, mask8, True, guide=mask4 if guid[1] else None, partial=part[1], upsample_nearest=True, bilinear_guided=bil[1], shared=sh[1])
    y = decoder2_block(p, cat(y, x4s), 8, mask4, True, guide=mask2 if guid[2] else None, partial=part[2], upsample_nearest=True, bilinear_guided=bil[2], shared=sh[2])
    y = decoder2_block(p, cat(y, x2s), 9, mask2, True, guide=mask if guid[3] else None, partial=part[3], upsample_nearest=True, bilinear_guided=bil[3], shared=sh[3])
    y = decoder2_block(p, cat(y, img), 10, mask, True, partial=part[4], shared=sh[4])
    vertex = conv2d(y, p["pv_final_conv_vertex.kernel"])
    out = np.concatenate([logits, vertex], 3)
    if return_intermediates:
        return out, dict(x2s=x2s, x4s=x4s, x8s=x8s, x32s=x32s, logits=logits, mask=mask)
    return out


# --------------------------------------------------------------------------------------
#  keypoint voting
# --------------------------------------------------------------------------------------


TF_PINV_RCOND = 10.0 * 2 * np.finfo(np.float64).eps   # tf.linalg.pinv(a, rcond=None) on a [2,2] fp64 matrix


def softplus(x):
    return np.logaddexp(0.0, x)


def label_components_4(binary: np.ndarray) -> np.ndarray:
    """4-connected component labelling of a 2-D 0/1 image, 0 = background, ids >= 1 --
    the published behaviour of tfa.image.connected_components (voting_layers_2d.py:51-56;
    SURVEY B7).  Ids are assigned in raster order of each component's first pixel."""
    h, w = binary.shape
    lab = np.zeros((h, w), dtype=np.int32)
    nxt = 0
    for y0 in range(h):
        for x0 in range(w):
            if binary[y0, x0] and lab[y0, x0] == 0:
                nxt += 1
                lab[y0, x0] = nxt
                stack = [(y0, x0)]
                while stack:
                    y, x = stack.pop()
                    for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                        yy, xx = y + dy, x + dx
                        if 0 <= yy < h and 0 <= xx < w and binary[yy, xx] and lab[yy, xx] == 0:
                            lab[yy, xx] = nxt
                            stack.append((yy, xx))
    return lab


def largest_component_filter(hot: np.ndarray, min_size: int = 50, rank: int = 1) -> np.ndarray:
    """voting_layers_2d.py:43-79 for one [h,w] 0/1 map: keep the component that ranks
    second in the size histogram (rank 0 is assumed to be id 0 = background) after zeroing
    bins with fewer than ``min_size`` pixels.  If that bin is empty the selected id is
    whatever top_k returns for a zero count: the lowest id among zero bins, which removes
    every foreground pixel unless that id is 0 (then background pixels would be selected,
    but they are multiplied by hot == 0 afterwards)."""
    comp = label_components_4(hot > 0)
    counts = np.bincount(comp.ravel(), minlength=rank + 1)   # rank 2 = output_second_largest_component: three bins, third entry (:58-59,71-73)
    counts = np.where(counts < min_size, 0, counts)
    # tf.math.top_k: descending, ties -> lower index first
    order = np.lexsort((np.arange(counts.size), -counts))
    keep_id = order[rank]
    return ((comp == keep_id) & (hot > 0)).astype(hot.dtype)


def ls_voting(seg, direct, conf, num_points=9, filter_estimates=False, hot_override=None, sigmoid_weights=False, second_largest=False):
    """CoordLSVotingWeighted.call/calc (voting_layers_2d.py:28-122).
    seg [B,H,W,K] logits, direct [B,H,W,2*kp] (dy,dx pairs), conf [B,H,W,kp].
    Returns [B,K-1,kp,2] keypoints in (y,x) pixels; accumulation in fp64."""
    b, h, w, k = seg.shape
    if sigmoid_weights:  # voting_layers_2d.py:32-33 with sigmoid_scale = 1
        wgt = (1.0 / (1.0 + np.exp(-conf.astype(np.float64)))).astype(np.float32)
    else:
        wgt = softplus(conf.astype(np.float32)).astype(np.float32)
    hot = saturated_softmax(seg.astype(np.float32))[..., 1:]
    if hot_override is not None:
        hot = hot_override
    if filter_estimates:
        hot_i = (hot + 0.1).astype(np.int32)
        keep = np.zeros_like(hot)
        for bi in range(b):
            for o in range(k - 1):
                keep[bi, :, :, o] = largest_component_filter(hot_i[bi, :, :, o], rank=2 if second_largest else 1)
        hot = keep * hot
    n = direct.reshape(b, h, w, num_points, 2).astype(np.float32)
    nrm = np.sqrt((n * n).sum(-1, keepdims=True))
    n = np.where(nrm > 0, n / np.where(nrm > 0, nrm, 1.0), 0.0).astype(np.float32)
    ny, nx = n[..., 0], n[..., 1]
    wv = wgt.reshape(b, h, w, num_points)
    r00 = (1.0 - ny * ny) * wv
    r01 = (-ny * nx) * wv
    r11 = (1.0 - nx * nx) * wv
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    cy = ((yy.astype(np.float32) + 0.5) / np.float32(h)).astype(np.float32)[None, :, :, None]
    cx = ((xx.astype(np.float32) + 0.5) / np.float32(h)).astype(np.float32)[None, :, :, None]
    q0 = r00 * cy + r01 * cx
    q1 = r01 * cy + r11 * cx
    out = np.zeros((b, k - 1, num_points, 2), dtype=np.float32)
    hot64 = hot.astype(np.float64)
    for o in range(k - 1):
        m = hot64[..., o][..., None]  # [b,h,w,1]
        s00 = (r00.astype(np.float64) * m).sum(axis=(1, 2))
        s01 = (r01.astype(np.float64) * m).sum(axis=(1, 2))
        s11 = (r11.astype(np.float64) * m).sum(axis=(1, 2))
        t0 = (q0.astype(np.float64) * m).sum(axis=(1, 2))
        t1 = (q1.astype(np.float64) * m).sum(axis=(1, 2))
        for bi in range(b):
            for j in range(num_points):
                mat = np.array([[s00[bi, j], s01[bi, j]], [s01[bi, j], s11[bi, j]]])
                # tf.linalg.pinv's documented default cut-off, 10 * max(rows, cols) * eps of the dtype (fp64 here: voting_layers_2d.py:111-116),
                # NOT NumPy's 1e-15: a direction field that is parallel to one part in 3e15 is rank 1 for TensorFlow
                sol = np.linalg.pinv(mat, rcond=TF_PINV_RCOND).dot(np.array([t0[bi, j], t1[bi, j]]))
                out[bi, o, j] = (sol * h).astype(np.float32)
    return out


def ls_voting_sums(seg, direct, conf, num_points=9):
    """The five fp64 accumulators per (b, object, keypoint) that precede the 2x2 solve:
    [S00, S01, S11, T0, T1] (voting_layers_2d.py:107-114)."""
    b, h, w, k = seg.shape
    lab = np.argmax(seg, axis=-1)
    wgt = softplus(conf.astype(np.float32)).astype(np.float32).reshape(b, h, w, num_points)
    n = direct.reshape(b, h, w, num_points, 2).astype(np.float32)
    nrm = np.sqrt((n * n).sum(-1, keepdims=True))
    n = np.where(nrm > 0, n / np.where(nrm > 0, nrm, 1.0), 0.0).astype(np.float32)
    ny, nx = n[..., 0], n[..., 1]
    r00 = (1.0 - ny * ny) * wgt
    r01 = (-ny * nx) * wgt
    r11 = (1.0 - nx * nx) * wgt
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    cy = ((yy.astype(np.float32) + 0.5) / np.float32(h)).astype(np.float32)[None, :, :, None]
    cx = ((xx.astype(np.float32) + 0.5) / np.float32(h)).astype(np.float32)[None, :, :, None]
    q0 = (r00 * cy + r01 * cx).astype(np.float64)  # per-pixel terms in fp32 like the reference
    q1 = (r01 * cy + r11 * cx).astype(np.float64)
    r00, r01, r11 = r00.astype(np.float64), r01.astype(np.float64), r11.astype(np.float64)
    sums = np.zeros((b, k - 1, num_points, 5))
    for o in range(k - 1):
        m = (lab == o + 1)[..., None]
        for i, a in enumerate((r00, r01, r11, q0, q1)):
            sums[:, o, :, i] = (a * m).sum(axis=(1, 2))
    return sums


# ---- RANSAC voting (ransac_voting.py:197-368) -----------------------------------------


def ransac_generate_hypothesis(direct, coords, idxs):
    """generate_hypothesis (ransac_voting.py:197-227).  direct [tn,vn,2] (x,y),
    coords [tn,2] (x,y), idxs [hn,vn,2] -> [hn,vn,2]; zero where |det| <= 1e-6."""
    hn, vn, _ = idxs.shape
    vi = np.arange(vn)[None, :, None]
    c_s = coords[idxs]  # [hn,vn,2,2]
    d_s = direct[idxs, vi]  # [hn,vn,2,2]
    det = d_s[:, :, 1, 0] * d_s[:, :, 0, 1] - d_s[:, :, 1, 1] * d_s[:, :, 0, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (
            (c_s[:, :, 1, 1] - c_s[:, :, 0, 1]) * d_s[:, :, 1, 0] - (c_s[:, :, 1, 0] - c_s[:, :, 0, 0]) * d_s[:, :, 1, 1]
        ) / det
        pts = c_s[:, :, 0] + d_s[:, :, 0] * u[..., None]
    return np.where((np.abs(det) > 1e-6)[..., None], pts, 0.0).astype(direct.dtype)


def ransac_vote(direct, coords, hyp, thresh):
    """voting_for_hypothesis (ransac_voting.py:230-249): inlier[h,t,v] in {0,1}."""
    hd = hyp[:, None, :, :] - coords[None, :, None, :]  # [hn,tn,vn,2]
    nd = np.sqrt((direct * direct).sum(-1))[None]  # [1,tn,vn]
    nh = np.sqrt((hd * hd).sum(-1))
    valid = (nd > 1e-6) & (nh > 1e-6) & (np.abs(hyp.sum(-1)) > 1e-6)[:, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        ang = (direct[None] * hd).sum(-1) / (nd * nh)
    return (valid & (ang > thresh)).astype(np.int32)


def ransac_voting_single(
    mask_hw,
    vertex_hw,
    idx_rounds: Sequence[np.ndarray],
    inlier_thresh=0.99,
    confidence=0.99,
    max_iter=20,
    min_num=5,
    max_num=30000,
    dtype=np.float32,
):
    """ransac_voting_batch (ransac_voting.py:276-368) for one object mask [h,w] (0/1) and
    vertex field [h,w,vn,2] in (y,x) order.  ``idx_rounds[r]`` is the [hn,vn,2] int array of
    pixel-pair indices the reference would draw with tf.random.uniform in round r
    (:319-321) -- injected so the run is reproducible.  Sub-sampling above ``max_num``
    (:295-301) is random in the reference and must be applied by the caller.
    Returns ([vn,2] (x,y), rounds_used)."""
    vn = vertex_hw.shape[2]
    fg = int((mask_hw != 0).sum())
    if fg < min_num:
        return np.zeros((vn, 2), dtype), 0
    ys, xs = np.nonzero(mask_hw)
    coords = np.stack([xs, ys], axis=1).astype(dtype) + dtype(0.5)
    direct = vertex_hw[ys, xs][:, :, ::-1].astype(dtype)  # -> (dx,dy)
    tn = coords.shape[0]
    win_ratio = np.zeros(vn, dtype)
    win_pts = np.zeros((vn, 2), dtype)
    hyp_num = 0.0
    it = 0
    while True:
        idxs = idx_rounds[it]
        hyp = ransac_generate_hypothesis(direct, coords, idxs)
        inl = ransac_vote(direct, coords, hyp, dtype(inlier_thresh))
        counts = inl.sum(axis=1)  # [hn,vn]
        widx = counts.argmax(axis=0)
        wcnt = counts.max(axis=0)
        wpts = hyp[widx, np.arange(vn)]
        ratio = wcnt.astype(dtype) / dtype(tn)
        larger = win_ratio < ratio
        win_pts = np.where(larger[:, None], wpts, win_pts)
        win_ratio = np.where(larger, ratio, win_ratio)
        hyp_num += idxs.shape[0]
        it += 1
        mn = float(win_ratio.min())
        if (1.0 - (1.0 - mn**2) ** hyp_num) > confidence or it >= max_iter:
            break
    normal = (direct * np.array([1, -1], dtype))[:, :, ::-1]  # (-dy, dx)
    inl = ransac_vote(direct, coords, win_pts[None], dtype(inlier_thresh))[0].astype(dtype)  # [tn,vn]
    normal = normal * inl[:, :, None]
    normal = np.transpose(normal, (1, 0, 2))  # [vn,tn,2]
    bvec = (normal * coords[None]).sum(2)
    ata = np.einsum("vti,vtj->vij", normal.astype(np.float64), normal.astype(np.float64)).astype(dtype)
    atb = (normal * bvec[:, :, None]).sum(1)
    sv = np.linalg.svd(ata.astype(np.float64), compute_uv=False)
    with np.errstate(divide="ignore", invalid="ignore"):
        cond = sv[:, 0] / sv[:, -1]
    if not np.all(np.isfinite(cond) & (cond < 1e6)):
        return win_pts.astype(dtype), it
    sol = np.linalg.solve(ata.astype(np.float64), atb.astype(np.float64)[..., None])[..., 0]
    return sol.astype(dtype), it


# --------------------------------------------------------------------------------------
#  synthetic inputs shared by tests / bench
# --------------------------------------------------------------------------------------


def synthetic_voting_inputs(b, h, w, num_obj=8, kp=9, seed=1237, noise=0.05, dtype=np.float32):
    """Seg logits from non-overlapping ellipses, exact unit vector field toward random
    keypoints + angular noise N(0, noise rad), conf logits N(0,1)."""
    rng = np.random.default_rng(seed)
    labels = np.zeros((b, h, w), dtype=np.int64)
    kps = np.zeros((b, num_obj, kp, 2), dtype=np.float64)  # (y,x)
    cols = int(math.ceil(math.sqrt(num_obj)))
    rows = int(math.ceil(num_obj / cols))
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    for bi in range(b):
        for o in range(num_obj):
            cy = (o // cols + 0.5) * h / rows
            cx = (o % cols + 0.5) * w / cols
            ry = rng.uniform(0.25, 0.45) * h / rows
            rx = rng.uniform(0.25, 0.45) * w / cols
            inside = ((yy + 0.5 - cy) / ry) ** 2 + ((xx + 0.5 - cx) / rx) ** 2 <= 1.0
            labels[bi][inside] = o + 1
            kps[bi, o, 0] = (cy, cx)
            kps[bi, o, 1:, 0] = rng.uniform(cy - ry, cy + ry, kp - 1)
            kps[bi, o, 1:, 1] = rng.uniform(cx - rx, cx + rx, kp - 1)
    seg = rng.standard_normal((b, h, w, num_obj + 1)).astype(dtype) * 0.1
    seg += 4.0 * onehot_from_labels(labels, num_obj + 1, dtype)
    direct = np.zeros((b, h, w, kp, 2), dtype=np.float64)
    for bi in range(b):
        for o in range(num_obj):
            m = labels[bi] == o + 1
            dy = kps[bi, o, :, 0][None, :] - (yy[m] + 0.5)[:, None]
            dx = kps[bi, o, :, 1][None, :] - (xx[m] + 0.5)[:, None]
            ang = np.arctan2(dy, dx) + noise * rng.standard_normal(dy.shape)
            direct[bi][m] = np.stack([np.sin(ang), np.cos(ang)], axis=-1)
    bg = labels == 0
    direct[bg] = 0.1 * rng.standard_normal((int(bg.sum()), kp, 2))
    conf = rng.standard_normal((b, h, w, kp)).astype(dtype)
    return seg, direct.reshape(b, h, w, kp * 2).astype(dtype), conf, labels, kps
