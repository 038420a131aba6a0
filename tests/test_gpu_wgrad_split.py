"""GPU parity of the bf16-pipe weight gradient (csrc/conv_wgrad_split.hip, cp_conv2d_wgrad_split) through the C ABI:
against an fp64 NumPy evaluation of  dW[ky,kx,ci,co] = sum_p X[p + tap][ci] * m(p, tap) * dY[p][co]  (what TensorFlow's
Conv2DBackpropFilter computes for the reference, train_casapose.py:594-611; m = the partial convolution's label mask,
_normalization_layers.py:333-371) pushed through the host packer, and against the fp32-MFMA kernel it replaces."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


def blob_labels(b, h, w, k, seed):
    rng = np.random.default_rng(seed)
    lab = np.zeros((b, h, w), np.uint8)
    for n in range(b):
        for c in range(1, k):
            y0, x0 = rng.integers(0, h - h // 3), rng.integers(0, w - w // 3)
            lab[n, y0:y0 + rng.integers(h // 4, h // 2), x0:x0 + rng.integers(w // 4, w // 2)] = c
    return lab


CASES = [
    # name, sources [(padded, real)], cout, partial, (b, h, w)
    ("64_64_two_strips", [(64, 64)], 64, False, (2, 20, 70)),
    ("two_sources_cout32_three_strips", [(64, 64), (32, 32)], 32, False, (1, 9, 130)),
    ("image_skip_partial", [(32, 32), (4, 3)], 32, True, (2, 18, 66)),
    ("image_skip_wide_rows", [(32, 32), (4, 3)], 32, False, (1, 11, 150)),
    ("image_skip_two_tiles", [(64, 64), (4, 3)], 64, False, (1, 9, 40)),   # image columns through the fp32 kernel (more than one 32x32 tile)
    ("128_128_four_tiles_row_chunks", [(128, 128)], 128, False, (1, 40, 33)),
    ("32_to_64", [(32, 32)], 64, False, (2, 7, 64)),
    ("96_odd_blocks_partial", [(96, 96)], 64, True, (2, 12, 40)),
    ("wide_partial_256_to_128", [(256, 256)], 128, True, (1, 14, 14)),
    ("one_row_one_column", [(32, 32)], 32, False, (3, 1, 1)),
]


def reference_hwio(xs, sources, dy, lab):
    b, h, w, _ = xs[0].shape
    x = np.concatenate([x_[..., :cr] for x_, (cp, cr) in zip(xs, sources)], axis=3).astype(np.float64)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    lp = None if lab is None else np.pad(lab.astype(np.int64) + 1, ((0, 0), (1, 1), (1, 1)))
    g = np.zeros((3, 3, x.shape[3], dy.shape[3]))
    for ky in range(3):
        for kx in range(3):
            xt = xp[:, ky:ky + h, kx:kx + w, :]
            if lp is not None:
                xt = xt * (lp[:, ky:ky + h, kx:kx + w] == (lab.astype(np.int64) + 1))[..., None]
            g[ky, kx] = np.einsum("bhwi,bhwo->io", xt, dy.astype(np.float64))
    return g


@pytest.mark.parametrize("planes", [3, 1, 0x12])   # 0x12 = CP_PLANES_F16X2: fp16 pairs, operands inside fp16's band as they are (round 6)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wgrad_split_matches_fp64_and_fp32_kernel(device, case, planes):
    from casapose_amd import _lib
    from casapose_amd._lib import ConvDesc, check

    lib = _lib.load()
    name, sources, cout, partial, (b, h, w) = case
    rng = np.random.default_rng(sum(name.encode()) + planes)
    xs = []
    for cp, cr in sources:
        x = np.zeros((b, h, w, cp), np.float32)
        x[..., :cr] = rng.standard_normal((b, h, w, cr)) * rng.uniform(0.5, 2.0)
        xs.append(x)
    ldo = cout + 32   # a dY row stride larger than cout
    dy = np.zeros((b, h, w, ldo), np.float32)
    dy[..., :cout] = rng.standard_normal((b, h, w, cout))
    lab = blob_labels(b, max(h, 3), max(w, 3), 4, 7)[:, :h, :w].copy() if partial else None
    xt = [torch.from_numpy(x).to(device) for x in xs]
    dyt = torch.from_numpy(dy).to(device)
    labt = torch.from_numpy(lab).to(device) if partial else None
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.cout = b, h, w, h, w, cout
    d.kh = d.kw = 3
    d.stride, d.dilation, d.pad = 1, 1, 1
    d.num_sources = len(sources)
    for s, (cp, cr) in enumerate(sources):
        d.src[s].data = xt[s].data_ptr()
        d.src[s].channels, d.src[s].ld, d.src[s].mode = cp, cp, 0
    d.tap_label = labt.data_ptr() if partial else None
    assert lib.cp_conv_wgrad_split_applicable(C.byref(d)) == 1
    chans = (C.c_int * 2)(*[s[0] for s in sources], *([0] * (2 - len(sources))))
    real = (C.c_int * 2)(*[s[1] for s in sources], *([0] * (2 - len(sources))))
    ktot = lib.cp_conv_ktot(3, 3, len(sources), chans)
    stream = torch.cuda.current_stream(device).cuda_stream
    got = torch.full((cout, ktot), 7.0, device=device)   # accumulate = 0 must overwrite
    check(lib.cp_conv2d_wgrad_split(C.byref(d), dyt.data_ptr(), ldo, got.data_ptr(), 0, planes, stream), "cp_conv2d_wgrad_split")
    f32 = torch.empty((cout, ktot), device=device)
    check(lib.cp_conv2d_wgrad_f32(C.byref(d), dyt.data_ptr(), ldo, f32.data_ptr(), 0, stream), "cp_conv2d_wgrad_f32")
    torch.cuda.synchronize()
    g = reference_hwio(xs, sources, dy[..., :cout], lab).astype(np.float32)
    ref = np.zeros((cout, ktot), np.float32)
    check(lib.cp_conv_pack_weights_host(g.ctypes.data, 0, 3, 3, cout, len(sources), chans, real, ref.ctypes.data), "cp_conv_pack_weights_host")
    tol = 2e-2 if planes == 1 else 3e-5
    got_h = got.cpu().numpy()
    # padding columns of the packed rows (image channel 3, the K padding up to a multiple of 32) carry no gradient
    used = ref != 0
    assert rel(got_h * used, ref) < tol, "against fp64"
    assert rel(f32.cpu().numpy() * used, ref) < 3e-5
    if planes != 1:
        assert rel(got_h * used, ref) <= 4 * max(rel(f32.cpu().numpy() * used, ref), 1e-6), "no worse than the fp32 MFMA kernel"
    # accumulate = 1 adds onto the previous content
    check(lib.cp_conv2d_wgrad_split(C.byref(d), dyt.data_ptr(), ldo, got.data_ptr(), 1, planes, stream), "cp_conv2d_wgrad_split")
    torch.cuda.synchronize()
    assert rel(got.cpu().numpy() * used, 2.0 * ref) < tol * 1.5


def test_wgrad_split_refuses_what_it_does_not_cover(device):
    from casapose_amd import _lib
    from casapose_amd._lib import ConvDesc

    lib = _lib.load()
    x = torch.zeros(1, 8, 8, 32, device=device)
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.cout = 1, 8, 8, 8, 8, 32
    d.kh = d.kw = 3
    d.stride, d.dilation, d.pad = 1, 2, 2
    d.num_sources = 1
    d.src[0].data, d.src[0].channels, d.src[0].ld = x.data_ptr(), 32, 32
    assert lib.cp_conv_wgrad_split_applicable(C.byref(d)) == 0   # dilation
    d.dilation, d.pad, d.cout = 1, 1, 9
    assert lib.cp_conv_wgrad_split_applicable(C.byref(d)) == 0   # cout not a multiple of 32
    out = torch.zeros(32 * 288, device=device)
    rc = lib.cp_conv2d_wgrad_split(C.byref(d), x.data_ptr(), 32, out.data_ptr(), 0, 3, None)
    assert rc != 0 and b"not covered" in lib.cp_last_error()


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("groups,rows,n,k", [(36, 128, 128, 128), (36, 384, 256, 384), (5, 200, 128, 256), (2, 2176, 512, 512), (3, 33, 384, 128)])
def test_wino_weight_gradient_gemm_on_the_bf16_pipe(device, planes, groups, rows, n, k):
    """cp_wino_wgrad_split_f32: du[g] = dm[g]^T v[g] (the grouped GEMM of the Winograd layers' weight gradient) against fp64 -- exact split at
    the fp32 gate and no worse than the fp32-MFMA grouped GEMM it replaces, bf16 operands at 2e-2; ragged row counts (the slab is 32 rows),
    several tiles per plane, stale content of du overwritten."""
    from casapose_amd import _lib
    from casapose_amd._lib import ConvDesc, check

    lib = _lib.load()
    rng = np.random.default_rng(groups + rows + n + k)
    dm = (rng.standard_normal((groups, rows, n)) * np.exp(rng.uniform(-3, 3, (1, 1, n)))).astype(np.float32)
    v = rng.standard_normal((groups, rows, k)).astype(np.float32)
    ref = np.einsum("gtn,gtk->gnk", dm.astype(np.float64), v.astype(np.float64))
    dmt, vt = torch.from_numpy(dm).to(device), torch.from_numpy(v).to(device)
    du = torch.full((groups, n, k), 3.0, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    assert lib.cp_wino_wgrad_split_applicable(groups, rows, n, k) == 1 and lib.cp_wino_wgrad_split_applicable(groups, rows, n + 32, k) == 0
    check(lib.cp_wino_wgrad_split_f32(dmt.data_ptr(), vt.data_ptr(), du.data_ptr(), groups, rows, n, k, planes, stream), "cp_wino_wgrad_split_f32")
    torch.cuda.synchronize()
    got = du.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    err = (np.abs(got - ref) / scale).max()
    assert err < (1e-5 if planes == 3 else 2e-2), err
    if planes == 3 and rows % 128 == 0:   # the fp32 grouped GEMM needs group_rows % 128 == 0
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w = 1, 1, groups * rows, 1, groups * rows
        d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = n, 1, 1, 1, 1, 0
        d.num_sources = 1
        d.src[0].data, d.src[0].channels, d.src[0].ld, d.src[0].mode = vt.data_ptr(), k, k, 0
        d.group_rows = rows
        f32 = torch.empty((groups, n, k), device=device)
        check(lib.cp_conv2d_wgrad_f32(C.byref(d), dmt.data_ptr(), n, f32.data_ptr(), 0, stream), "cp_conv2d_wgrad_f32(grouped)")
        e32 = (np.abs(f32.cpu().numpy().astype(np.float64) - ref) / scale).max()
        assert err <= 2.0 * e32 + 1e-7, (err, e32)


@pytest.mark.parametrize("groups,rows,n,k,mag", [(36, 384, 256, 384, 3e-5), (5, 200, 128, 256, 40.0), (2, 2176, 512, 512, 1e-5)])
def test_wino_weight_gradient_gemm_in_the_fp16_two_way_split(device, groups, rows, n, k, mag):
    """cp_wino_wgrad_split_scaled_f32 with CP_PLANES_F16X2: a gradient-like operand of arbitrary magnitude is brought into fp16's band by the power of
    two cp_f16x2_range_check derives from its maximum (what the training plan reads from the monitor slot of cp_wino_dy_transform_f32); error
    against fp64 no worse than the exact split's gate and within 2x of the fp32-MFMA grouped GEMM."""
    from casapose_amd import _lib
    from casapose_amd._lib import check

    lib = _lib.load()
    rng = np.random.default_rng(groups + rows + n + k)
    dm = (rng.standard_normal((groups, rows, n)) * np.exp(rng.uniform(-3, 3, (1, 1, n))) * mag).astype(np.float32)
    v = (rng.standard_normal((groups, rows, k)) * 3.0).astype(np.float32)
    ref = np.einsum("gtn,gtk->gnk", dm.astype(np.float64), v.astype(np.float64))
    dmt, vt = torch.from_numpy(dm).to(device), torch.from_numpy(v).to(device)
    stream = torch.cuda.current_stream(device).cuda_stream
    rescale = C.c_float(0.0)
    assert lib.cp_f16x2_range_check(float(np.abs(dm).max()), 0.5, 65504.0 / 4, C.byref(rescale)) in (0, 1)
    du = torch.full((groups, n, k), 3.0, device=device)
    check(lib.cp_wino_wgrad_split_scaled_f32(dmt.data_ptr(), vt.data_ptr(), du.data_ptr(), groups, rows, n, k, _lib.PLANES_F16X2, rescale.value, 1.0, stream))
    ex = torch.empty((groups, n, k), device=device)
    check(lib.cp_wino_wgrad_split_f32(dmt.data_ptr(), vt.data_ptr(), ex.data_ptr(), groups, rows, n, k, 3, stream))
    torch.cuda.synchronize()
    scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
    err = (np.abs(du.cpu().numpy().astype(np.float64) - ref) / scale).max()
    err3 = (np.abs(ex.cpu().numpy().astype(np.float64) - ref) / scale).max()
    assert err < 1e-5 and err <= 4.0 * err3 + 2e-7, (err, err3)
    # factors other than 1 belong to the fp16 form only
    assert lib.cp_wino_wgrad_split_scaled_f32(dmt.data_ptr(), vt.data_ptr(), du.data_ptr(), groups, rows, n, k, 3, 2.0, 1.0, stream) != 0
