"""GPU parity of the LDS-resident halo-tile 3x3 kernel (conv_halo.hip, CP_TILE_HALO) against the
oracle AND against the generic implicit-GEMM kernel on the same inputs, for every operand mode it
covers and for ragged tile edges."""
import numpy as np
import pytest
import torch

import casapose_oracle as O
from test_gpu_conv import _labels, close, dev

pytestmark = pytest.mark.gpu
HALO, GENERIC = 7, 5


@pytest.mark.parametrize("cin,cout,hw", [(32, 32, (8, 32)), (64, 32, (13, 45)), (64, 64, (21, 70)), (128, 48, (6, 33)), (96, 12, (4, 31))])
def test_plain_and_ragged(device, cin, cout, hw):
    from casapose_amd import ops

    rng = np.random.default_rng(cin + cout + hw[0])
    x = rng.standard_normal((2, hw[0], hw[1], cin))
    w = rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    ref = O.conv2d(x, w, pad=1)
    raw, _ = ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1, tile_hint=HALO)
    close(raw, ref)
    gen, _ = ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1, tile_hint=GENERIC)
    close(raw, gen.cpu().numpy().astype(np.float64), rtol=2e-6)


def test_auto_dispatch_picks_halo(device):
    from casapose_amd import _lib
    from casapose_amd.engine import FusedConv

    w = np.zeros((3, 3, 64, 32), np.float32)
    layer = FusedConv("t", w, 0, 3, 3, 32, [(64, 64)], device)
    x = torch.zeros(1, 8, 32, 64, device=device)
    o = torch.empty(1, 8, 32, 32, device=device)
    layer.bind(batch=1, in_h=8, in_w=32, pad=1, srcs=[dict(data=x, ld=64)], out_raw=o)
    assert _lib.load().cp_conv_selected_tile(layer.desc) == HALO
    layer.bind(batch=1, in_h=8, in_w=32, pad=2, dilation=2, srcs=[dict(data=x, ld=64)], out_raw=o)
    assert _lib.load().cp_conv_selected_tile(layer.desc) != HALO


def test_two_sources_residual_dual_output(device):
    from casapose_amd import ops

    rng = np.random.default_rng(6)
    a, b = rng.standard_normal((2, 14, 40, 64)), rng.standard_normal((2, 14, 40, 32))
    w = rng.standard_normal((3, 3, 96, 64)) / 30.0
    res = rng.standard_normal((2, 14, 40, 64))
    sc, sh = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.2
    raw_ref = O.conv2d(np.concatenate([a, b], 3), w, pad=1) + res
    raw, act = ops.conv2d_fused([dev(a, device), dev(b, device)], w.astype(np.float32), pad=1, residual=dev(res, device),
                                scale=dev(sc, device), shift=dev(sh, device), act=2, want_raw=True, want_act=True, tile_hint=HALO)
    close(raw, raw_ref)
    close(act, O.leaky_as_relu_pair(raw_ref * sc + sh))


def test_feature_plus_image_and_bilinear(device):
    from casapose_amd import ops

    rng = np.random.default_rng(7)
    f = rng.standard_normal((2, 16, 40, 32))
    img = rng.uniform(-1, 1, (2, 16, 40, 3))
    w = rng.standard_normal((3, 3, 35, 32)) / 18.0
    img4 = ops.pad_channels_3to4(dev(img, device))
    raw, _ = ops.conv2d_fused([dev(f, device), img4], w.astype(np.float32), pad=1, real_channels=[32, 3], tile_hint=HALO)
    close(raw, O.conv2d(np.concatenate([f, img], 3), w, pad=1))
    # bilinear x2 source + image (decoder block 5)
    low = rng.standard_normal((2, 8, 20, 32))
    raw2, _ = ops.conv2d_fused([dev(low, device), img4], w.astype(np.float32), pad=1, real_channels=[32, 3], modes=[2, 0], tile_hint=HALO)
    close(raw2, O.conv2d(np.concatenate([O.upsample_bilinear_x2(low), img], 3), w, pad=1))
    # bilinear x2 source + 64-channel skip (decoder block 4)
    low2, skip = rng.standard_normal((1, 10, 24, 64)), rng.standard_normal((1, 20, 48, 64))
    w2 = rng.standard_normal((3, 3, 128, 32)) / 34.0
    raw3, _ = ops.conv2d_fused([dev(low2, device), dev(skip, device)], w2.astype(np.float32), pad=1, modes=[2, 0], tile_hint=HALO)
    close(raw3, O.conv2d(np.concatenate([O.upsample_bilinear_x2(low2), skip], 3), w2, pad=1))


def test_halo_needs_four_channel_groups(device):
    """the halo kernel's epilogue moves four channels per 16-byte access: cout % 4 != 0 goes to the generic kernel, and forcing
    CP_TILE_HALO for such a layer is an error, not a silent switch."""
    from casapose_amd import _lib, ops

    rng = np.random.default_rng(5)
    x, w = rng.standard_normal((1, 4, 31, 96)), rng.standard_normal((3, 3, 96, 9)) * 0.1
    with pytest.raises(_lib.CasaposeHipError):
        ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1, tile_hint=HALO)
    raw, _ = ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1)
    close(raw, O.conv2d(x, w, pad=1))


@pytest.mark.parametrize("hw", [(24, 32), (22, 50)])
def test_partial_clade_and_guided(device, hw):
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
    # partial conv + CLADE + leaky on a direct 64-channel source
    x = rng.standard_normal((b, h, w, 64))
    wt = rng.standard_normal((64, 3, 3, 32)) / 24.0
    y = O.clade_weighted(O.partial_convolution(x, wt, mask), mask, p["c.gamma"], p["c.beta"], p["c.moving_mean"], p["c.moving_variance"])
    _, act = ops.conv2d_fused([dev(x, device)], wt.astype(np.float32), layout=1, pad=1, tap_label=labels[0], row_scale=pnorm[0],
                              scale=dev(ts, device), shift=dev(tb, device), epi_label=labels[0], act=2, want_raw=False, want_act=True,
                              tile_hint=HALO)
    close(act, O.leaky_as_relu_pair(y))
    # guided x2 source + image, partial conv (decoder block 10)
    low = rng.standard_normal((b, h // 2, w // 2, 32))
    img = rng.uniform(-1, 1, (b, h, w, 3))
    wt2 = rng.standard_normal((35, 3, 3, 32)) / 18.0
    up = O.guided_upsampling(low, O.half_size(mask), mask)
    ref = O.partial_convolution(np.concatenate([up, img], 3), wt2, mask)
    raw, _ = ops.conv2d_fused([dev(low, device), ops.pad_channels_3to4(dev(img, device))], wt2.astype(np.float32), layout=1, pad=1,
                              real_channels=[32, 3], modes=[1, 0], sels=[sel[0], None], tap_label=labels[0], row_scale=pnorm[0],
                              tile_hint=HALO)
    close(raw, ref)
