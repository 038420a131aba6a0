"""GPU parity of the bf16-pipe 3x3 kernel (csrc/conv_hsplit.hip) against the fp64 oracle:
  planes = 3 (exact three-way bf16 split, six products): the SAME gate as the fp32-MFMA kernels (1e-4 of the tensor's range; measured
             error is at the fp32 round-off level), over 12 orders of magnitude of operand scale;
  planes = 1 (operands rounded to bf16): the bf16 gate of SURVEY 8(d), 3e-2 relative;
  F16X2 (fp16 two-way split, three products; round 4): the fp32 gate again -- its error against the fp32 kernels' is pinned in test_gpu_f16x2.py.
Every operand / epilogue mode the kernel covers: one and two sources, the 4-channel image source, ragged tiles (16-row x 32-column
tiles), partial-convolution tap mask with 9/count, CLADE table + leaky pair, residual, dual outputs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import casapose_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from test_gpu_conv import _labels, close, dev

pytestmark = pytest.mark.gpu
SPLIT3, BF16, F16X2 = 100, 101, 102
MODES = [(SPLIT3, 1e-4), (BF16, 3e-2), (F16X2, 1e-4)]   # F16X2: the fp16 two-way split (round 4), the fp32 gate


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("cin,cout,hw", [(16, 32, (16, 32)), (32, 32, (8, 32)), (64, 32, (13, 45)), (64, 64, (21, 70)), (128, 48, (37, 33)), (96, 12, (4, 31)),
                                         (48, 64, (33, 65)), (64, 128, (19, 40)), (96, 256, (9, 33)), (32, 72, (12, 31))])
def test_plain_and_ragged(device, mode, tol, cin, cout, hw):
    from casapose_amd import ops

    rng = np.random.default_rng(cin + cout + hw[0])
    x = rng.standard_normal((2, hw[0], hw[1], cin))
    w = rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    raw, _ = ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1, tile_hint=mode)
    close(raw, O.conv2d(x, w, pad=1), rtol=tol)


def test_exact_split_is_fp32_equivalent_over_operand_scales(device):
    """the three-way split is exact for any finite fp32 operand, so the error stays at the fp32 level when the operands are scaled by
    1e-6 ... 1e6 and when they are ill-scaled against each other; it is compared with the error of the fp32-MFMA halo kernel."""
    from casapose_amd import ops

    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 32, 64, 64))
    w = rng.standard_normal((3, 3, 64, 64)) / 24.0
    for sx, sw in ((1.0, 1.0), (1e-6, 1.0), (1e6, 1e-6), (1e3, 1e3)):
        ref = O.conv2d(x * sx, w * sw, pad=1)
        scale = np.abs(ref).max()
        got = ops.conv2d_fused([dev(x * sx, device)], (w * sw).astype(np.float32), pad=1, tile_hint=SPLIT3)[0].cpu().numpy().astype(np.float64)
        f32 = ops.conv2d_fused([dev(x * sx, device)], (w * sw).astype(np.float32), pad=1, tile_hint=7)[0].cpu().numpy().astype(np.float64)
        ref32 = O.conv2d((x * sx).astype(np.float32).astype(np.float64), (w * sw).astype(np.float32).astype(np.float64), pad=1)
        e_split, e_f32 = np.abs(got - ref32).max() / scale, np.abs(f32 - ref32).max() / scale
        assert e_split < 2e-6 and e_split <= 2.0 * e_f32 + 2e-7, (sx, sw, e_split, e_f32)


@pytest.mark.parametrize("mode,tol", MODES)
def test_two_sources_residual_dual_output(device, mode, tol):
    from casapose_amd import ops

    rng = np.random.default_rng(6)
    a, b = rng.standard_normal((2, 30, 40, 64)), rng.standard_normal((2, 30, 40, 32))
    w = rng.standard_normal((3, 3, 96, 64)) / 30.0
    res = rng.standard_normal((2, 30, 40, 64))
    sc, sh = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.2
    raw_ref = O.conv2d(np.concatenate([a, b], 3), w, pad=1) + res
    raw, act = ops.conv2d_fused([dev(a, device), dev(b, device)], w.astype(np.float32), pad=1, residual=dev(res, device),
                                scale=dev(sc, device), shift=dev(sh, device), act=2, want_raw=True, want_act=True, tile_hint=mode)
    close(raw, raw_ref, rtol=tol)
    close(act, O.leaky_as_relu_pair(raw_ref * sc + sh), rtol=tol)


@pytest.mark.parametrize("mode,tol", MODES)
def test_feature_plus_image(device, mode, tol):
    from casapose_amd import ops

    rng = np.random.default_rng(7)
    f = rng.standard_normal((2, 35, 40, 32))
    img = rng.uniform(-1, 1, (2, 35, 40, 3))
    w = rng.standard_normal((3, 3, 35, 32)) / 18.0
    img4 = ops.pad_channels_3to4(dev(img, device))
    raw, _ = ops.conv2d_fused([dev(f, device), img4], w.astype(np.float32), pad=1, real_channels=[32, 3], tile_hint=mode)
    close(raw, O.conv2d(np.concatenate([f, img], 3), w, pad=1), rtol=tol)
    # only the image part non-zero / only the feature part non-zero: the two K segments are wired to the right weights
    w_img = w.copy()
    w_img[:, :, :32] = 0
    raw, _ = ops.conv2d_fused([dev(f, device), img4], w_img.astype(np.float32), pad=1, real_channels=[32, 3], tile_hint=mode)
    close(raw, O.conv2d(img, w[:, :, 32:], pad=1), rtol=tol)


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("hw", [(24, 32), (38, 50)])
def test_partial_clade(device, mode, tol, hw):
    from casapose_amd import ops
    from casapose_amd.engine import fold_clade

    rng = np.random.default_rng(8 + hw[1])
    b, (h, w), k = 2, hw, 5
    lab = _labels(rng, b, h, w, k)
    mask = O.onehot_from_labels(lab, k)
    labels, pnorm, sel = ops.label_pyramid(dev(lab, device, torch.uint8))
    p = {"c.gamma": rng.uniform(0.5, 1.5, (k, 32)), "c.beta": rng.standard_normal((k, 32)) * 0.2,
         "c.moving_mean": rng.standard_normal(32) * 0.1, "c.moving_variance": rng.uniform(0.5, 1.5, 32)}
    ts, tb = fold_clade(p, "c")
    x = rng.standard_normal((b, h, w, 64))
    wt = rng.standard_normal((64, 3, 3, 32)) / 24.0
    y = O.clade_weighted(O.partial_convolution(x, wt, mask), mask, p["c.gamma"], p["c.beta"], p["c.moving_mean"], p["c.moving_variance"])
    _, act = ops.conv2d_fused([dev(x, device)], wt.astype(np.float32), layout=1, pad=1, tap_label=labels[0], row_scale=pnorm[0],
                              scale=dev(ts, device), shift=dev(tb, device), epi_label=labels[0], act=2, want_raw=False, want_act=True, tile_hint=mode)
    close(act, O.leaky_as_relu_pair(y), rtol=tol)
    # feature + image, partial conv (decoder block 10 with a materialised upsampling)
    f = rng.standard_normal((b, h, w, 32))
    img = rng.uniform(-1, 1, (b, h, w, 3))
    wt2 = rng.standard_normal((35, 3, 3, 32)) / 18.0
    ref = O.partial_convolution(np.concatenate([f, img], 3), wt2, mask)
    raw, _ = ops.conv2d_fused([dev(f, device), ops.pad_channels_3to4(dev(img, device))], wt2.astype(np.float32), layout=1, pad=1,
                              real_channels=[32, 3], tap_label=labels[0], row_scale=pnorm[0], tile_hint=mode)
    close(raw, ref, rtol=tol)


@pytest.mark.parametrize("mode,tol", MODES)
def test_fused_upsampling_sources(device, mode, tol):
    """source 0 read through the x2 bilinear filter (decoder blocks 3-5) or through the guided-upsampling selection map (blocks 8-10)"""
    from casapose_amd import ops

    rng = np.random.default_rng(17)
    f = rng.standard_normal((2, 13, 20, 32))
    img = rng.uniform(-1, 1, (2, 26, 40, 3))
    w = rng.standard_normal((3, 3, 35, 32)) / 18.0
    img4 = ops.pad_channels_3to4(dev(img, device))
    raw, _ = ops.conv2d_fused([dev(f, device), img4], w.astype(np.float32), pad=1, real_channels=[32, 3], modes=[2, 0], tile_hint=mode)
    close(raw, O.conv2d(np.concatenate([O.upsample_bilinear_x2(f), img], 3), w, pad=1), rtol=tol)
    low2, skip = rng.standard_normal((1, 10, 24, 64)), rng.standard_normal((1, 20, 48, 64))
    w2 = rng.standard_normal((3, 3, 128, 64)) / 34.0
    raw3, _ = ops.conv2d_fused([dev(low2, device), dev(skip, device)], w2.astype(np.float32), pad=1, modes=[2, 0], tile_hint=mode)
    close(raw3, O.conv2d(np.concatenate([O.upsample_bilinear_x2(low2), skip], 3), w2, pad=1), rtol=tol)
    # guided x2 source + image / + skip, partial conv (decoder blocks 10 and 8)
    b, h, wd, k = 2, 24, 48, 5
    lab = _labels(rng, b, h, wd, k)
    mask = O.onehot_from_labels(lab, k)
    labels, pnorm, sel = ops.label_pyramid(dev(lab, device, torch.uint8))
    low = rng.standard_normal((b, h // 2, wd // 2, 32))
    imgp = rng.uniform(-1, 1, (b, h, wd, 3))
    wt = rng.standard_normal((35, 3, 3, 32)) / 18.0
    up = O.guided_upsampling(low, O.half_size(mask), mask)
    raw4, _ = ops.conv2d_fused([dev(low, device), ops.pad_channels_3to4(dev(imgp, device))], wt.astype(np.float32), layout=1, pad=1, real_channels=[32, 3],
                               modes=[1, 0], sels=[sel[0], None], tap_label=labels[0], row_scale=pnorm[0], tile_hint=mode)
    close(raw4, O.partial_convolution(np.concatenate([up, imgp], 3), wt, mask), rtol=tol)
    low64, skip64 = rng.standard_normal((b, h // 2, wd // 2, 64)), rng.standard_normal((b, h, wd, 64))
    wt2 = rng.standard_normal((128, 3, 3, 64)) / 34.0
    up64 = O.guided_upsampling(low64, O.half_size(mask), mask)
    raw5, _ = ops.conv2d_fused([dev(low64, device), dev(skip64, device)], wt2.astype(np.float32), layout=1, pad=1, modes=[1, 0], sels=[sel[0], None],
                               tap_label=labels[0], row_scale=pnorm[0], tile_hint=mode)
    close(raw5, O.partial_convolution(np.concatenate([up64, skip64], 3), wt2, mask), rtol=tol)


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("c0,c1,cout,hw", [(16, 0, 32, (144, 320)), (32, 4, 32, (136, 288)), (48, 32, 32, (72, 608)), (32, 16, 64, (152, 304)), (64, 64, 64, (40, 48))])
def test_bilinear_source_from_the_staged_low_resolution_tile(device, mode, tol, c0, c1, cout, hw):
    """Round 4: the loaders stage the half-resolution tile of source 0 in LDS and interpolate the halo from it.  Shapes with MORE tiles than
    blocks (a block walks several tiles: the three slice cursors cross tile borders), one-slice tiles (16 channels: every slice starts a
    tile), bilinear slices followed by direct slices / by the image block, ragged right and bottom edges, both cout widths."""
    from casapose_amd import ops

    rng = np.random.default_rng(c0 + c1 + cout)
    b, (h, w) = 2, hw
    low = rng.standard_normal((b, h // 2, w // 2, c0))
    srcs, real, full = [dev(low, device)], [c0], [O.upsample_bilinear_x2(low)]
    if c1 == 4:
        img = rng.uniform(-1, 1, (b, h, w, 3))
        srcs.append(ops.pad_channels_3to4(dev(img, device)))
        real.append(3)
        full.append(img)
    elif c1:
        skip = rng.standard_normal((b, h, w, c1))
        srcs.append(dev(skip, device))
        real.append(c1)
        full.append(skip)
    cin = sum(real)
    wk = rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    raw, _ = ops.conv2d_fused(srcs, wk.astype(np.float32), pad=1, real_channels=real, modes=[2] + [0] * (len(srcs) - 1), tile_hint=mode)
    close(raw, O.conv2d(np.concatenate(full, 3), wk, pad=1), rtol=tol)


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("q", [9, 27, 32])
def test_fused_head(device, mode, tol, q):
    """the 1x1 head (pv_final_conv_segmentation / _vertex) fused into the epilogue of a 32-channel layer, on the same matrix pipe"""
    from casapose_amd import _lib
    from casapose_amd.engine import FusedConv

    rng = np.random.default_rng(23 + q)
    b, h, w = 2, 19, 45
    f, img = rng.standard_normal((b, h, w, 32)), rng.uniform(-1, 1, (b, h, w, 3))
    wk = rng.standard_normal((3, 3, 35, 32)) / 18.0
    wh = rng.standard_normal((1, 1, 32, q)) / 6.0
    sc, sh = rng.uniform(0.5, 1.5, 32), rng.standard_normal(32) * 0.2
    from casapose_amd import ops

    layer = FusedConv("t", wk.astype(np.float32), 0, 3, 3, 32, [(32, 32), (4, 3)], device)
    layer.attach_head(wh.astype(np.float32))
    out = torch.zeros(b, h, w, 36, device=device)
    layer.bind(batch=b, in_h=h, in_w=w, pad=1, srcs=[dict(data=dev(f, device), ld=32), dict(data=ops.pad_channels_3to4(dev(img, device)), ld=4)],
               scale=dev(sc, device), shift=dev(sh, device), act=2, head_out=out, head_out_ld=36, tile_hint=mode)
    layer.run(torch.cuda.current_stream(device).cuda_stream)
    act = O.leaky_as_relu_pair(O.conv2d(np.concatenate([f, img], 3), wk, pad=1) * sc + sh)
    ref = O.conv2d(act, wh)
    got = out.cpu().numpy().astype(np.float64)
    assert np.abs(got[..., :q] - ref).max() <= tol * np.abs(ref).max()
    assert (got[..., q:] == 0).all()        # channels beyond the head stay untouched


@pytest.mark.parametrize("mode,tol", MODES)
def test_wide_partial_layer_in_passes(device, mode, tol):
    """more than 64 output channels = several passes of 64 over the tile list (decoder blocks 6 / 7: 512 -> 256, 384 -> 128, partial)"""
    from casapose_amd import ops
    from casapose_amd.engine import fold_clade

    rng = np.random.default_rng(41)
    b, h, w, k, cin, cout = 2, 15, 20, 4, 128, 160
    lab = _labels(rng, b, h, w, k)
    mask = O.onehot_from_labels(lab, k)
    labels, pnorm, _ = ops.label_pyramid(dev(lab, device, torch.uint8))
    p = {"c.gamma": rng.uniform(0.5, 1.5, (k, cout)), "c.beta": rng.standard_normal((k, cout)) * 0.2,
         "c.moving_mean": rng.standard_normal(cout) * 0.1, "c.moving_variance": rng.uniform(0.5, 1.5, cout)}
    ts, tb = fold_clade(p, "c")
    x = rng.standard_normal((b, h, w, cin))
    wt = rng.standard_normal((cin, 3, 3, cout)) / 34.0
    y = O.clade_weighted(O.partial_convolution(x, wt, mask), mask, p["c.gamma"], p["c.beta"], p["c.moving_mean"], p["c.moving_variance"])
    raw, act = ops.conv2d_fused([dev(x, device)], wt.astype(np.float32), layout=1, pad=1, tap_label=labels[0], row_scale=pnorm[0],
                                scale=dev(ts, device), shift=dev(tb, device), epi_label=labels[0], act=1, want_raw=True, want_act=True, tile_hint=mode)
    close(raw, O.partial_convolution(x, wt, mask), rtol=tol)
    close(act, O.relu(y), rtol=tol)


def test_out_of_range_layers_are_refused(device):
    from casapose_amd import _lib, ops

    rng = np.random.default_rng(5)
    x, w = rng.standard_normal((1, 8, 32, 64)), rng.standard_normal((3, 3, 64, 520)) * 0.1
    with pytest.raises(_lib.CasaposeHipError):
        ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1, tile_hint=SPLIT3)        # cout > 512
    w2 = rng.standard_normal((3, 3, 64, 32)) * 0.1
    with pytest.raises(_lib.CasaposeHipError):
        ops.conv2d_fused([dev(x, device)], w2.astype(np.float32), pad=2, dilation=2, tile_hint=SPLIT3)   # dilated


@pytest.mark.parametrize("planes,tol", [(3, 1e-4), (1, 3e-2), (0x12, 1e-4)])
@pytest.mark.parametrize("hw", [(64, 96), (37, 70), (480, 640)])
def test_stem_on_the_bf16_pipe(device, planes, tol, hw):
    """csrc/conv_stem_split.hip (round 4): conv0 -- 7x7 / stride 2 / pad 3, bn_data as an affine on the real pixels (the padding stays zero),
    bn0 + ReLU epilogue, raw and activated outputs -- as exact three-way bf16 splits (the fp32 gate) or with bf16 operands, against the fp64
    oracle; odd sizes (ragged tiles, halo rows beyond the image), and the full-size map (more tiles than blocks)."""
    from casapose_amd import _lib
    from casapose_amd.engine import FusedConv

    lib = _lib.load()
    rng = np.random.default_rng(hw[0] + planes)
    b, (h, w) = (1 if hw[0] > 100 else 2), hw
    img = rng.uniform(-1, 1, (b, h, w, 3))
    wk = rng.standard_normal((7, 7, 3, 64)) / np.sqrt(147.0)
    ps, pb = rng.uniform(0.5, 1.5, 3), rng.standard_normal(3) * 0.3
    sc, sh = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.2
    layer = FusedConv("conv0", wk.astype(np.float32), 0, 7, 7, 64, [(4, 3)], device)
    from casapose_amd import ops

    img4 = ops.pad_channels_3to4(dev(img, device))
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    raw, act = torch.empty(b, oh, ow, 64, device=device), torch.empty(b, oh, ow, 64, device=device)
    pre = (dev(np.r_[ps, 1.0], device), dev(np.r_[pb, 0.0], device))
    layer.bind(batch=b, in_h=h, in_w=w, stride=2, pad=3, srcs=[dict(data=img4, ld=4, pre=pre)], scale=dev(sc, device), shift=dev(sh, device),
               act=_lib.ACT_RELU, out_raw=raw, out_act=act)
    assert lib.cp_conv_selected_tile(C.byref(layer.desc)) == _lib.TILE_STEM
    layer.stem_split = planes
    layer.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = O.conv2d(img * ps + pb, wk, stride=2, pad=3)   # zero padding AFTER the affine: O.conv2d pads the affine image with zeros
    close(raw, ref, rtol=tol)
    close(act, np.maximum(ref * sc + sh, 0.0), rtol=tol)
    if planes in (3, 0x12):   # and it is the fp32-MFMA stem kernel's result to fp32 rounding
        layer.stem_split = 0
        raw2 = torch.empty_like(raw)
        layer.desc.out_raw = raw2.data_ptr()
        layer.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert float((raw - raw2).abs().max()) <= 2e-5 * float(raw2.abs().max())


def test_head_layers_epilogue_split_is_bit_identical_to_the_unsplit_form(device):
    """Round 5 (`epi_split`, csrc/conv_hsplit.hip): in the head layers a consumer wave hands the accumulators of its second row to the loader wave
    w + 4, which runs the SAME epilogue code one tile later.  Nothing about the arithmetic changes, so the whole forward (logits, vector field, the
    label map the fused head writes) must be BIT-identical with CASAPOSE_HS_EPI_SPLIT=0 and =1 -- at a size with several tiles per block, partial
    tiles on the right / bottom edge and an odd number of tiles, in f16x2 and in the bf16 mode (the two arithmetics whose LDS budget admits the split).
    The switch is read once per process: two child processes."""
    import hashlib
    import os
    import subprocess
    import sys

    code = r'''
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
out = []
for mode in ("f16x2", "bf16"):
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=9, input_shape=(104, 200, 3), weights=None, base_model="resnet18", device=dev, seed=5, conv_mode=mode)
    img = (2 * torch.rand(3, 104, 200, 3, generator=torch.Generator().manual_seed(2)) - 1).to(dev)
    y = net([img], training=False)
    lab = net._net.plan(3, 104, 200).labels[0]
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    out.append(hashlib.sha256(y.cpu().numpy().tobytes() + lab.cpu().numpy().tobytes()).hexdigest())
print(" ".join(out))
''' % (ROOT, os.path.join(ROOT, "oracle"))
    digests = {}
    for split in ("0", "1"):
        env = dict(os.environ, CASAPOSE_HS_EPI_SPLIT=split)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[split] = r.stdout.strip().splitlines()[-1]
    assert digests["0"] == digests["1"], digests
