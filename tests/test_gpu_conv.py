"""GPU parity: the fused implicit-GEMM convolution (cp_conv2d_fwd_f32, through the C ABI)
against the fp64 NumPy oracle, on seeded inputs small enough for the oracle to finish in
seconds.  Tolerance: fp32 MFMA accumulation vs fp64 -> max|diff| <= 1e-4 * max|ref|
(indexing mistakes produce O(1) relative errors)."""
import ctypes as C

import numpy as np
import pytest
import torch

import casapose_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def dev(a, device, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


def close(got: torch.Tensor, ref: np.ndarray, rtol=RTOL):
    g = got.detach().cpu().numpy().astype(np.float64)
    assert g.shape == ref.shape, (g.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-6)
    err = np.abs(g - ref).max()
    assert err <= rtol * scale, "max|diff| %.3e vs scale %.3e" % (err, scale)


@pytest.mark.parametrize(
    "cin,cout,k,stride,dil,pad,hw",
    [
        (32, 64, 3, 1, 1, 1, (20, 28)),
        (64, 128, 3, 2, 1, 1, (21, 30)),
        (64, 64, 3, 1, 2, 2, (18, 22)),
        (128, 160, 3, 1, 4, 4, (16, 24)),
        (64, 96, 1, 2, 1, 0, (22, 26)),
        (32, 9, 1, 1, 1, 0, (24, 40)),
        (96, 27, 1, 1, 1, 0, (9, 11)),
    ],
)
def test_plain_conv(device, cin, cout, k, stride, dil, pad, hw):
    from casapose_amd import ops

    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = rng.standard_normal((2, hw[0], hw[1], cin))
    w = rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)
    ref = O.conv2d(x, w, stride=stride, dilation=dil, pad=pad)
    raw, _ = ops.conv2d_fused([dev(x, device)], w.astype(np.float32), stride=stride, dilation=dil, pad=pad)
    close(raw, ref)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6])
def test_every_tile_shape(device, tile):
    from casapose_amd import ops

    rng = np.random.default_rng(tile)
    x = rng.standard_normal((3, 19, 23, 64))
    w = rng.standard_normal((3, 3, 64, 72)) / 24.0
    ref = O.conv2d(x, w, pad=1)
    raw, _ = ops.conv2d_fused([dev(x, device)], w.astype(np.float32), pad=1, tile_hint=tile)
    close(raw, ref)


def test_conv0_c4_source_with_input_affine(device):
    """conv0 path: 3-channel image padded to 4, bn_data as the operand affine applied BEFORE the
    zero padding (resnet.py:247-249), 7x7 stride 2 pad 3, bn0+relu epilogue."""
    from casapose_amd import ops

    rng = np.random.default_rng(5)
    img = rng.uniform(-1, 1, (2, 30, 44, 3))
    w = rng.standard_normal((7, 7, 3, 64)) / 12.0
    ps, pb = rng.uniform(0.5, 1.5, 3), rng.standard_normal(3) * 0.3
    es, eb = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.3
    ref = O.relu(O.conv2d(img * ps + pb, w, stride=2, pad=3) * es + eb)
    img4 = ops.pad_channels_3to4(dev(img, device))
    pre = (dev(np.append(ps, 0.0), device), dev(np.append(pb, 0.0), device))
    _, act = ops.conv2d_fused([img4], w.astype(np.float32), stride=2, pad=3, pre=[pre], real_channels=[3],
                              scale=dev(es, device), shift=dev(eb, device), act=1, want_raw=False, want_act=True)
    close(act, ref)


@pytest.mark.parametrize("hw", [(30, 44), (64, 96), (38, 130)])
def test_stem_kernel_matches_oracle_and_generic_route(device, hw):
    """csrc/conv_stem.hip (the 7x7 / stride 2 / pad 3 stem with the input halo resident in LDS) is what an automatic launch picks for
    conv0; it must agree with the fp64 oracle and with the implicit-GEMM kernel (tile hint) on ragged sizes, raw and activated outputs,
    with and without the input affine."""
    from casapose_amd import _lib, ops

    rng = np.random.default_rng(hw[0])
    img = rng.uniform(-1, 1, (3, hw[0], hw[1], 3))
    w = rng.standard_normal((7, 7, 3, 64)) / 12.0
    ps, pb = rng.uniform(0.5, 1.5, 3), rng.standard_normal(3) * 0.3
    es, eb = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.3
    raw_ref = O.conv2d(img * ps + pb, w, stride=2, pad=3)
    img4 = ops.pad_channels_3to4(dev(img, device))
    pre = (dev(np.append(ps, 0.0), device), dev(np.append(pb, 0.0), device))
    kw = dict(stride=2, pad=3, pre=[pre], real_channels=[3], scale=dev(es, device), shift=dev(eb, device), act=1, want_raw=True, want_act=True)
    raw, act = ops.conv2d_fused([img4], w.astype(np.float32), tile_hint=_lib.TILE_STEM, **kw)
    close(raw, raw_ref)
    close(act, O.relu(raw_ref * es + eb))
    raw_g, act_g = ops.conv2d_fused([img4], w.astype(np.float32), tile_hint=_lib.TILE_64x64, **kw)
    close(raw, raw_g.cpu().numpy().astype(np.float64), rtol=2e-6)
    close(act, act_g.cpu().numpy().astype(np.float64), rtol=2e-6)
    plain, _ = ops.conv2d_fused([img4], w.astype(np.float32), stride=2, pad=3, real_channels=[3])  # automatic: the stem kernel, no affine
    close(plain, O.conv2d(img, w, stride=2, pad=3))


def test_two_sources_residual_dual_output(device):
    from casapose_amd import ops

    rng = np.random.default_rng(6)
    a = rng.standard_normal((2, 14, 18, 64))
    b = rng.standard_normal((2, 14, 18, 32))
    w = rng.standard_normal((3, 3, 96, 64)) / 30.0
    res = rng.standard_normal((2, 14, 18, 64))
    sc, sh = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.2
    raw_ref = O.conv2d(np.concatenate([a, b], 3), w, pad=1) + res
    act_ref = O.leaky_as_relu_pair(raw_ref * sc + sh)
    raw, act = ops.conv2d_fused([dev(a, device), dev(b, device)], w.astype(np.float32), pad=1, residual=dev(res, device),
                                scale=dev(sc, device), shift=dev(sh, device), act=2, want_raw=True, want_act=True)
    close(raw, raw_ref)
    close(act, act_ref)


def test_feature_plus_image_source(device):
    """blocks 5/10: concat[32-channel feature, raw 3-channel image] (pose_models.py:545,607)."""
    from casapose_amd import ops

    rng = np.random.default_rng(7)
    f = rng.standard_normal((2, 16, 20, 32))
    img = rng.uniform(-1, 1, (2, 16, 20, 3))
    w = rng.standard_normal((3, 3, 35, 32)) / 18.0
    ref = O.conv2d(np.concatenate([f, img], 3), w, pad=1)
    img4 = ops.pad_channels_3to4(dev(img, device))
    raw, _ = ops.conv2d_fused([dev(f, device), img4], w.astype(np.float32), pad=1, real_channels=[32, 3])
    close(raw, ref)


def _labels(rng, b, h, w, k):
    lab = np.zeros((b, h, w), dtype=np.int64)
    for bi in range(b):
        for o in range(1, k):
            y0, x0 = rng.integers(0, h - 4), rng.integers(0, w - 4)
            lab[bi, y0 : y0 + rng.integers(3, h // 2 + 2), x0 : x0 + rng.integers(3, w // 2 + 2)] = o
    return lab


def test_partial_conv_with_clade_epilogue(device):
    """PartialConvolution + ClassAdaptiveWeightedNormalization + leaky pair
    (_normalization_layers.py:325-373,119-139; casapose.py:98-105) in ONE launch."""
    from casapose_amd import ops
    from casapose_amd.engine import fold_clade

    rng = np.random.default_rng(8)
    b, h, w, k, cin, cout = 2, 24, 32, 5, 64, 32
    lab = _labels(rng, b, h, w, k)
    mask = O.onehot_from_labels(lab, k)
    x = rng.standard_normal((b, h, w, cin))
    wt = rng.standard_normal((cin, 3, 3, cout)) / 24.0
    p = {
        "c.gamma": rng.uniform(0.5, 1.5, (k, cout)),
        "c.beta": rng.standard_normal((k, cout)) * 0.2,
        "c.moving_mean": rng.standard_normal(cout) * 0.1,
        "c.moving_variance": rng.uniform(0.5, 1.5, cout),
    }
    y = O.partial_convolution(x, wt, mask)
    y = O.clade_weighted(y, mask, p["c.gamma"], p["c.beta"], p["c.moving_mean"], p["c.moving_variance"])
    ref = O.leaky_as_relu_pair(y)
    labels, pnorm, _ = ops.label_pyramid(dev(lab, device, torch.uint8))
    ts, tb = fold_clade(p, "c")
    _, act = ops.conv2d_fused([dev(x, device)], wt.astype(np.float32), layout=1, pad=1, tap_label=labels[0], row_scale=pnorm[0],
                              scale=dev(ts, device), shift=dev(tb, device), epi_label=labels[0], act=2, want_raw=False, want_act=True)
    close(act, ref)


def test_partial_conv_all_one_label_is_border_rescaled_conv(device):
    """KAT: with a single label everywhere the partial conv equals the ordinary
    SAME conv scaled by 9 / (#in-bounds taps)."""
    from casapose_amd import ops

    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, 10, 12, 32))
    wt = rng.standard_normal((32, 3, 3, 32)) / 17.0
    lab = np.zeros((1, 10, 12), np.uint8)
    labels, pnorm, _ = ops.label_pyramid(dev(lab, device, torch.uint8))
    raw, _ = ops.conv2d_fused([dev(x, device)], wt.astype(np.float32), layout=1, pad=1, tap_label=labels[0], row_scale=pnorm[0])
    plain = O.conv2d(x, np.transpose(wt, (1, 2, 0, 3)), pad=1)
    cnt = O.conv2d(np.ones((1, 10, 12, 1)), np.ones((3, 3, 1, 1)), pad=1)
    close(raw, plain * 9.0 / cnt)


def test_fused_guided_and_bilinear_sources(device):
    """decoder blocks 3-5 / 8-10: the x2 upsampling of the previous block is fused into the operand
    gather of the consumer (GuidedUpsampling, _normalization_layers.py:507-566; UpSampling2D bilinear,
    casapose.py:135-140)."""
    from casapose_amd import ops

    rng = np.random.default_rng(10)
    b, h, w, k = 2, 16, 24, 4
    lab = _labels(rng, b, h, w, k)
    mask = O.onehot_from_labels(lab, k)
    mask_lo = O.half_size(mask)
    low = rng.standard_normal((b, h // 2, w // 2, 64))
    skip = rng.standard_normal((b, h, w, 32))
    wt = rng.standard_normal((96, 3, 3, 32)) / 29.0
    labels, pnorm, sel = ops.label_pyramid(dev(lab, device, torch.uint8))
    # guided + partial conv
    up = O.guided_upsampling(low, mask_lo, mask)
    ref = O.partial_convolution(np.concatenate([up, skip], 3), wt, mask)
    raw, _ = ops.conv2d_fused([dev(low, device), dev(skip, device)], wt.astype(np.float32), layout=1, pad=1, modes=[1, 0],
                              sels=[sel[0], None], tap_label=labels[0], row_scale=pnorm[0])
    close(raw, ref)
    # stand-alone guided upsampling kernel agrees too
    close(ops.guided_upsample_x2(dev(low, device), sel[0]), up, rtol=1e-7)
    # bilinear + plain conv
    w2 = np.transpose(wt, (1, 2, 0, 3))
    upb = O.upsample_bilinear_x2(low)
    ref2 = O.conv2d(np.concatenate([upb, skip], 3), w2, pad=1)
    raw2, _ = ops.conv2d_fused([dev(low, device), dev(skip, device)], w2.astype(np.float32), pad=1, modes=[2, 0])
    close(raw2, ref2)
    close(ops.upsample_bilinear_x2(dev(low, device)), upb, rtol=1e-6)


def test_aux_kernels(device):
    from casapose_amd import ops

    rng = np.random.default_rng(11)
    x = np.maximum(rng.standard_normal((2, 15, 22, 64)), 0)  # post-ReLU like relu0
    sc, sh = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64) * 0.2
    close(ops.maxpool3x3s2(dev(x, device)), O.maxpool_3x3_s2_pad1(x), rtol=1e-7)
    close(ops.maxpool3x3s2(dev(x, device), dev(sc, device), dev(sh, device), relu=True), O.relu(O.maxpool_3x3_s2_pad1(x) * sc + sh), rtol=1e-6)
    # negative inputs: the zero padding must win at the border (resnet.py:253)
    xn = -np.abs(rng.standard_normal((1, 6, 6, 4))) - 1.0
    close(ops.maxpool3x3s2(dev(xn, device)), O.maxpool_3x3_s2_pad1(xn), rtol=1e-7)
    logits = rng.standard_normal((2, 12, 16, 12))
    got = ops.argmax_labels(dev(logits, device), classes=9, offset=0).cpu().numpy()
    assert (got == logits[..., :9].argmax(-1)).all()
    lab = _labels(rng, 2, 32, 48, 6)
    labels, pnorm, sel = ops.label_pyramid(dev(lab, device, torch.uint8))
    masks = [O.onehot_from_labels(lab, 6)]
    for _ in range(3):
        masks.append(O.half_size(masks[-1]))
    for l in range(4):
        assert (labels[l].cpu().numpy() == masks[l].argmax(-1)).all()
        _, norm = O.partial_conv_mask(masks[l])
        close(pnorm[l], norm[..., 0], rtol=1e-6)
    for l in range(3):
        assert (sel[l].cpu().numpy() == O.guided_upsampling_select(masks[l + 1], masks[l])).all()


def test_bad_arguments_report_errors(device):
    from casapose_amd import _lib, ops

    x = torch.zeros(1, 8, 8, 24, device=device)  # 24 channels: neither 4 nor a multiple of 32
    with pytest.raises((_lib.CasaposeHipError, ValueError)):
        ops.conv2d_fused([x], np.zeros((3, 3, 24, 8), np.float32), pad=1)


@pytest.mark.parametrize(
    "srcs,cout,dil,hw,epi",
    [
        ([256], 128, 1, (13, 18), "plain"),
        ([256], 256, 2, (15, 20), "residual_relu"),
        ([512], 160, 4, (15, 20), "dual"),
        ([256, 128], 128, 1, (12, 16), "clade_leaky"),
        ([256], 132, 4, (9, 7), "plain"),           # sub-grids smaller than one tile, ragged cout
    ],
)
def test_winograd_conv(device, srcs, cout, dil, hw, epi):
    """Winograd F(4x4,3x3) path (csrc/wino.hip + grouped GEMM) vs the fp64 oracle of the convolution it replaces:
    dilation by sub-grid decomposition, concatenated sources, ragged tiles, every epilogue flavour."""
    from casapose_amd import _lib
    from casapose_amd.engine import WinoConv

    b = 2
    rng = np.random.default_rng(sum(srcs) + cout + dil)
    xs = [rng.standard_normal((b, hw[0], hw[1], c)) for c in srcs]
    cin = sum(srcs)
    w = rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)
    ref = O.conv2d(np.concatenate(xs, 3), w, dilation=dil, pad=dil)
    layer = WinoConv("t", w.astype(np.float32), cout, [(c, c) for c in srcs], device)
    _, tp = WinoConv.tiles(b, hw[0], hw[1], dil)
    V = torch.empty(36 * tp * cin, device=device)
    M = torch.empty(36 * tp * cout, device=device)
    kw = {}
    res = lab = None
    if epi in ("residual_relu", "dual"):
        res = rng.standard_normal(ref.shape)
        kw["residual"] = dev(res, device)
        ref = ref + res
    raw = torch.zeros(b, hw[0], hw[1], cout, device=device)
    act = torch.zeros(b, hw[0], hw[1], cout, device=device)
    ref_act = None
    if epi == "plain":
        kw["out_raw"] = raw
    elif epi == "residual_relu":
        sc, sh = rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.2
        kw.update(scale=dev(sc, device), shift=dev(sh, device), act=_lib.ACT_RELU, out_act=act)
        ref_act = O.relu(ref * sc + sh)
    elif epi == "dual":
        sc, sh = rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.2
        kw.update(scale=dev(sc, device), shift=dev(sh, device), act=_lib.ACT_RELU, out_act=act, out_raw=raw)
        ref_act = O.relu(ref * sc + sh)
    else:
        k = 5
        lab = rng.integers(0, k, (b, hw[0], hw[1]))
        sc, sh = rng.uniform(0.5, 1.5, (k, cout)), rng.standard_normal((k, cout)) * 0.2
        kw.update(scale=dev(sc, device), shift=dev(sh, device), epi_label=dev(lab, device, torch.uint8), act=_lib.ACT_LEAKY01, out_act=act)
        ref_act = O.leaky_as_relu_pair(ref * sc[lab] + sh[lab])
    layer.bind(batch=b, in_h=hw[0], in_w=hw[1], dilation=dil, srcs=[dict(data=dev(x, device), ld=x.shape[3]) for x in xs], V=V, M=M, **kw)
    layer.run(torch.cuda.current_stream(device).cuda_stream)
    torch.cuda.synchronize()
    if "out_raw" in kw:
        close(raw, ref)
    if ref_act is not None:
        close(act, ref_act)


def test_split_bf16_gemm_is_fp32_equivalent(device):
    """cp_wino_gemm_split_f32 (opt-in; csrc/wino_gemm_split.hip): the grouped GEMM with every operand split exactly into three bf16 terms
    and six products accumulated in fp32.  Against an fp64 product its error must not exceed the exact-fp32 MFMA kernel's by more than
    rounding noise, on well-scaled data and on data spanning 12 orders of magnitude (the split is exact for any exponent)."""
    from casapose_amd import _lib
    from casapose_amd._lib import check
    from casapose_amd.engine import split_wino_weights

    lib = _lib.load()
    st = torch.cuda.current_stream(device).cuda_stream
    g = torch.Generator().manual_seed(5)
    for rows, group, n, k, spread in [(256, 128, 128, 64, 0.0), (3 * 384, 384, 256, 96, 6.0), (2 * 128, 128, 132, 32, 0.0)]:
        V = torch.randn(rows, k, generator=g) * torch.pow(10.0, spread * (torch.rand(rows, 1, generator=g) - 0.5))
        U = torch.randn(rows // group, n, k, generator=g) * torch.pow(10.0, spread * (torch.rand(rows // group, n, 1, generator=g) - 0.5))
        Vd, Ud = V.to(device), U.to(device)
        M1, M2 = torch.empty(rows, n, device=device), torch.empty(rows, n, device=device)
        check(lib.cp_wino_gemm_f32(Vd.data_ptr(), Ud.data_ptr(), M1.data_ptr(), rows, group, k, n, st), "fp32")
        Us = split_wino_weights(Ud, rows // group, n, k)
        check(lib.cp_wino_gemm_split_f32(Vd.data_ptr(), Us.data_ptr(), M2.data_ptr(), rows, group, k, n, st), "split")
        ref = torch.cat([V[i * group:(i + 1) * group].double() @ U[i].double().T for i in range(rows // group)])
        scale = (V.double().abs() @ torch.ones(k, 1, dtype=torch.float64)).clamp_min(1e-300)  # row-wise magnitude: errors are relative to sum |v||u|
        bound = torch.cat([V[i * group:(i + 1) * group].double().abs() @ U[i].double().abs().T for i in range(rows // group)]).clamp_min(1e-300)
        e1 = float(((M1.cpu().double() - ref).abs() / bound).max())
        e2 = float(((M2.cpu().double() - ref).abs() / bound).max())
        assert e2 < 2e-6 and e2 < 2.0 * e1 + 1e-7, (rows, n, k, e1, e2)
        # two planes (hi + mid, three products; the bf16 conv modes): 16 significand bits per operand -> 2^-15 of sum |v||u|, far from fp32
        M3 = torch.empty(rows, n, device=device)
        check(lib.cp_wino_gemm_split_planes_f32(Vd.data_ptr(), Us.data_ptr(), M3.data_ptr(), rows, group, k, n, 2, st), "split, two planes")
        e3 = float(((M3.cpu().double() - ref).abs() / bound).max())
        assert 1e-7 < e3 < 4e-5, (rows, n, k, e3)
        del scale


@pytest.mark.parametrize("h,w,dil,cout", [(60, 80, 4, 512), (56, 56, 4, 256), (15, 20, 1, 64), (30, 37, 2, 96), (9, 7, 4, 32), (60, 80, 2, 64)])
@pytest.mark.parametrize("with_residual", [False, True])
def test_fused_winograd_output_input_transform_equals_the_two_passes(device, h, w, dil, cout, with_residual):
    """cp_wino_output_input_transform_f32 (round 4): output transform + epilogue of one Winograd layer and input transform of the next in one
    launch -- against cp_wino_output_transform_f32 followed by cp_wino_input_transform_f32 on the stored activated map.  The raw and the
    activated output must be BIT-EQUAL (same expressions); the transformed input V to the last bits (the compiler contracts the B^T d B sums
    into FMAs differently in the two kernels: <= 2e-6 of the largest |V|); ragged sub-grids (rows / columns beyond the image inside the last
    tiles) and the zero ring of the next convolution's padding included."""
    from casapose_amd import _lib

    lib = _lib.load()
    b = 2
    # (60, 80, 2): 8 x 10 tiles per sub-grid -> the 16-channel block shape, which the plan does not select (cp_wino_output_input_applicable says
    # no: it measured slower than the two passes) but the entry point still serves
    assert lib.cp_wino_output_input_applicable(b, h, w, dil, cout) == (0 if (h, w, dil) == (60, 80, 2) else 1)
    st = torch.cuda.current_stream().cuda_stream
    t, tp = C.c_int(0), C.c_int(0)
    _lib.check(lib.cp_wino_tiles(b, h, w, dil, C.byref(t), C.byref(tp)))
    g = torch.Generator(device="cpu").manual_seed(h * w + cout)
    M = torch.randn(36, tp.value, cout, generator=g).to(device)
    res = torch.randn(b, h, w, cout, generator=g).to(device) if with_residual else None
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(device), torch.randn(cout, generator=g).to(device)
    raw0, act0 = torch.empty(b, h, w, cout, device=device), torch.empty(b, h, w, cout, device=device)
    V0 = torch.full((36, tp.value, cout), float("nan"), device=device)
    rp = res.data_ptr() if res is not None else None
    _lib.check(lib.cp_wino_output_transform_f32(M.data_ptr(), cout, b, h, w, dil, rp, cout, scale.data_ptr(), shift.data_ptr(), None, 1, raw0.data_ptr(), cout,
                                                act0.data_ptr(), cout, st))
    _lib.check(lib.cp_wino_input_transform_f32(act0.data_ptr(), cout, cout, b, h, w, dil, V0.data_ptr(), cout, 0, st))
    raw1, act1 = torch.empty_like(raw0), torch.empty_like(act0)
    V1 = torch.full_like(V0, float("nan"))
    _lib.check(lib.cp_wino_output_input_transform_f32(M.data_ptr(), cout, b, h, w, dil, rp, cout, scale.data_ptr(), shift.data_ptr(), 1, raw1.data_ptr(), cout,
                                                      act1.data_ptr(), cout, V1.data_ptr(), cout, 0, st))
    torch.cuda.synchronize()
    T = t.value
    assert torch.equal(raw0, raw1) and torch.equal(act0, act1)
    vmax = float(V0[:, :T].abs().max())
    assert torch.isfinite(V1[:, :T]).all() and float((V0[:, :T] - V1[:, :T]).abs().max()) <= 2e-6 * vmax
    # without the optional stores, into a channel slice of a wider V
    V2 = torch.zeros(36, tp.value, cout + 32, device=device)
    _lib.check(lib.cp_wino_output_input_transform_f32(M.data_ptr(), cout, b, h, w, dil, rp, cout, scale.data_ptr(), shift.data_ptr(), 1, None, cout, None, cout,
                                                      V2.data_ptr(), cout + 32, 32, st))
    torch.cuda.synchronize()
    assert float((V2[:, :T, 32:] - V0[:, :T]).abs().max()) <= 2e-6 * vmax and not V2[:, :, :32].any()
