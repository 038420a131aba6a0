"""GPU parity of the TRAINING path (forward with batch statistics, losses, hand-written backward, Adam)
against the fp64 PyTorch-autograd restatement in oracle/torch_train_ref.py, through the C ABI."""
import ctypes as C

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import casapose_oracle as O
import torch_train_ref as R

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


def rel_l2(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)


def blob_labels(b, h, w, k, seed):
    rng = np.random.default_rng(seed)
    lab = np.zeros((b, h, w), np.uint8)
    for n in range(b):
        for c in range(1, k):
            y0, x0 = rng.integers(0, h - h // 3), rng.integers(0, w - w // 3)
            lab[n, y0:y0 + rng.integers(h // 4, h // 2), x0:x0 + rng.integers(w // 4, w // 2)] = c
    return lab


# --------------------------------------------------------------------------------------------------
# single-layer checks: weight and data gradient of every convolution flavour on the path
# --------------------------------------------------------------------------------------------------
CONV_CASES = [
    # name, k, stride, dil, pad, sources [(padded, real)], cout, partial, (h, w)
    ("3x3", 3, 1, 1, 1, [(64, 64)], 64, False, (12, 20)),
    ("3x3_two_sources", 3, 1, 1, 1, [(64, 64), (32, 32)], 32, False, (10, 12)),
    ("3x3_image_skip", 3, 1, 1, 1, [(32, 32), (4, 3)], 32, False, (16, 16)),
    ("3x3_dil2", 3, 1, 2, 2, [(32, 32)], 96, False, (9, 11)),
    ("3x3_dil4_wide", 3, 1, 4, 4, [(128, 128)], 160, False, (8, 8)),
    ("3x3_stride2", 3, 2, 1, 1, [(64, 64)], 128, False, (12, 16)),
    ("1x1_stride2", 1, 2, 1, 0, [(64, 64)], 128, False, (12, 16)),
    ("1x1_head", 1, 1, 1, 0, [(32, 32)], 9, False, (10, 14)),
    ("1x1_gemm_route", 1, 1, 1, 0, [(128, 128)], 256, False, (16, 16)),       # stage shortcut: bf16-pipe GEMMs for forward, data and weight gradient
    ("1x1_gemm_route_small", 1, 1, 1, 0, [(64, 64)], 64, False, (16, 24)),     # ... whose weight gradient stays on the fp32 kernel (64 < 128)
    ("7x7_stride2_image", 7, 2, 1, 3, [(4, 3)], 64, False, (16, 24)),
    ("3x3_winograd_dil2", 3, 1, 2, 2, [(256, 256)], 256, False, (15, 20)),
    ("3x3_winograd_two_sources", 3, 1, 1, 1, [(256, 256), (128, 128)], 128, False, (12, 16)),
    ("3x3_winograd_dil4_ragged", 3, 1, 4, 4, [(256, 256)], 384, False, (9, 7)),
    ("3x3_partial", 3, 1, 1, 1, [(64, 64)], 32, True, (12, 16)),
    ("3x3_partial_image", 3, 1, 1, 1, [(32, 32), (4, 3)], 32, True, (16, 16)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_wgrad_dgrad(device, case):
    from casapose_amd import _lib
    from casapose_amd.train_engine import ConvOp, ParamStore, TrainConv, TT

    name, k, stride, dil, pad, sources, cout, partial, (h, w) = case
    b = 2
    rng = np.random.default_rng(sum(name.encode()))
    cin = sum(s[1] for s in sources)
    wk = rng.standard_normal((k, k, cin, cout)).astype(np.float32) * 0.1
    store = ParamStore({"L.kernel": wk}, device)
    needs = [s[0] != 4 for s in sources]
    layer = TrainConv(store, "L.kernel", 0, k, cout, sources, needs)
    xs = []
    for (cp, cr), ng in zip(sources, needs):
        x = np.zeros((b, h, w, cp), np.float32)
        x[..., :cr] = rng.standard_normal((b, h, w, cr))
        xs.append(x)
    tts = [TT(torch.from_numpy(x).to(device), ng) for x, ng in zip(xs, needs)]
    eff = (k - 1) * dil + 1
    oh, ow = (h + 2 * pad - eff) // stride + 1, (w + 2 * pad - eff) // stride + 1
    ldo = (cout + 31) // 32 * 32
    out = TT(torch.zeros(b, oh, ow, ldo, device=device), True)
    lab_t = pn_t = None
    lab = None
    if partial:
        lab = blob_labels(b, h, w, 4, 5)
        lab_t = torch.from_numpy(lab).to(device)
        cnt = np.zeros((b, h, w))
        lp = np.pad(lab.astype(np.int64) + 1, ((0, 0), (1, 1), (1, 1)))
        for ky in range(3):
            for kx in range(3):
                cnt += lp[:, ky:ky + h, kx:kx + w] == (lab + 1)
        pn = (9.0 / cnt).astype(np.float32)
        pn_t = torch.from_numpy(pn).to(device)
    op = ConvOp(layer, [(t, t.c) for t in tts], (out.data, 0, ldo), b, h, w, stride=stride, dilation=dil, pad=pad, tap_label=lab_t, row_scale=pn_t,
                out=out, dy_ptr_ld=(out.grad, 0, ldo))
    stream = torch.cuda.current_stream(device).cuda_stream
    if "gemm_route" in name:  # what TrainPlan does for the 1x1 / stride-1 shortcuts
        from casapose_amd.train_engine import conv_split_planes

        op.setup_gemm()
        if conv_split_planes():   # CASAPOSE_CONV_MODE=f32 keeps these layers on the fp32-MFMA kernels
            assert op.gemm is not None and op.gemm["wgrad"] == ("small" not in name and os.environ.get("CASAPOSE_WINO_WGRAD", "split") != "f32")
        else:
            assert op.gemm is None
    if "winograd" in name:  # what TrainPlan does for the deep layers: Winograd forward + data gradient
        nv, nm = op.setup_winograd()
        assert nv > 0 and op.wino_fwd is not None and (len(op.wino_dgrad) >= 1 or cout < 256)  # the data gradient's K is cout
        op.bind_winograd(torch.empty(nv, device=device), torch.empty(nm, device=device))
    layer.refresh(stream)
    op.forward(stream)
    # reference (fp64 autograd)
    xr = [torch.tensor(x[..., :cr].astype(np.float64), requires_grad=True) for x, (cp, cr) in zip(xs, sources)]
    wr = torch.tensor(wk.astype(np.float64), requires_grad=True)
    xin = torch.cat(xr, dim=3)
    if partial:
        yr = R.partial_conv(xin, wr.permute(2, 0, 1, 3), torch.from_numpy(lab.astype(np.int64)))
    else:
        yr = R.conv_nhwc(xin, wr, stride=stride, dilation=dil, pad=pad)
    got = out.data.cpu().numpy()[..., :cout]
    assert rel(got, yr.detach().numpy()) < 2e-5
    dy = rng.standard_normal((b, oh, ow, cout))
    yr.backward(torch.from_numpy(dy))
    dyp = np.zeros((b, oh, ow, ldo), np.float32)
    dyp[..., :cout] = dy
    if partial:  # the gradient of the un-normalised sum, as the normalisation backward hands it over
        dyp *= pn[..., None]
    out.grad.copy_(torch.from_numpy(dyp))
    out.has_grad = True
    op.backward(stream)
    torch.cuda.synchronize()
    gw = store.grad_view("L.kernel").cpu().numpy()
    assert rel(gw, wr.grad.numpy()) < 3e-5, "weight gradient"
    for t, x64, ng in zip(tts, xr, needs):
        if ng:
            assert rel(t.grad.cpu().numpy(), x64.grad.numpy()) < 3e-5, "data gradient"
    # accumulation into an existing gradient (tensor with several consumers)
    for t in tts:
        if t.needs_grad:
            t.grad.fill_(1.0)
            t.has_grad = True
    op.backward(stream)
    torch.cuda.synchronize()
    for t, x64, ng in zip(tts, xr, needs):
        if ng:
            assert rel(t.grad.cpu().numpy(), x64.grad.numpy() + 1.0) < 3e-5, "accumulated data gradient"


# --------------------------------------------------------------------------------------------------
# normalisation + activation, resampling adjoints, loss, Adam
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("classes,act,c", [(1, 1, 64), (1, 2, 32), (5, 2, 32), (9, 1, 256), (1, 0, 4)])
def test_bn_act_forward_backward(device, hip_lib, classes, act, c):
    lib = hip_lib
    from casapose_amd._lib import check

    rng = np.random.default_rng(c + classes)
    b, h, w = 2, 12, 10
    n = b * h * w
    x = (rng.standard_normal((n, c)) * 2 + 0.5).astype(np.float32)
    lab = rng.integers(0, classes, n).astype(np.uint8)
    gamma = (1 + 0.3 * rng.standard_normal((classes, c))).astype(np.float32)
    beta = (0.2 * rng.standard_normal((classes, c))).astype(np.float32)
    dy = rng.standard_normal((n, c)).astype(np.float32)
    rs = rng.uniform(0.5, 2.0, n).astype(np.float32)
    st = torch.cuda.current_stream(device).cuda_stream
    keep = []

    def d(a):
        keep.append(torch.from_numpy(np.ascontiguousarray(a)).to(device))  # stays alive for the whole test
        return keep[-1]

    xd, labd, dyd = d(x), d(lab), d(dy)
    sums = torch.zeros(2 * c, dtype=torch.float64, device=device)
    check(lib.cp_bn_stats_f32(xd.data_ptr(), n, c, c, sums.data_ptr(), st))
    s = sums.cpu().numpy()
    x64 = x.astype(np.float64)
    assert rel(s[:c], x64.sum(0)) < 1e-12 and rel(s[c:], (x64 ** 2).sum(0)) < 1e-12
    # reference
    xt = torch.tensor(x64, requires_grad=True)
    gt, bt = torch.tensor(gamma.astype(np.float64), requires_grad=True), torch.tensor(beta.astype(np.float64), requires_grad=True)
    mean = xt.mean(0)
    var = ((xt - mean) ** 2).mean(0)
    xh = (xt - mean) / torch.sqrt(var + R.BN_EPS)
    li = torch.from_numpy(lab.astype(np.int64))
    t = gt[li] * xh + bt[li]
    yr = t if act == 0 else (F.relu(t) if act == 1 else R.leaky_pair(t))
    yr.backward(torch.from_numpy(dy.astype(np.float64)))
    # device forward
    meanv = (s[:c] / n)
    varv = s[c:] / n - meanv ** 2
    rstd = 1.0 / np.sqrt(varv + R.BN_EPS)
    scale = (gamma * rstd[None]).astype(np.float32)
    shift = (beta - gamma * (meanv * rstd)[None]).astype(np.float32)
    y = torch.empty(n, c, device=device)
    scd, shd = d(scale), d(shift)
    check(lib.cp_affine_act_f32(xd.data_ptr(), n, c, c, scd.data_ptr(), shd.data_ptr(), labd.data_ptr() if classes > 1 else None, act,
                                y.data_ptr(), c, st))
    assert rel(y.cpu().numpy(), yr.detach().numpy()) < 2e-5
    # device backward
    red = torch.zeros(classes * c * 2, dtype=torch.float64, device=device)
    chan = torch.zeros(c * 2, dtype=torch.float64, device=device)
    md, rd, gd, bd = d(meanv.astype(np.float32)), d(rstd.astype(np.float32)), d(gamma), d(beta)
    lp = labd.data_ptr() if classes > 1 else None
    # both forms of the branch decision: recomputed from gamma / beta (tables None), and the forward's own fma(x, scale, shift) (the training plan)
    for fs, fb in ((None, None), (scd.data_ptr(), shd.data_ptr())):
        check(lib.cp_bn_act_bwd_reduce_f32(xd.data_ptr(), c, dyd.data_ptr(), c, n, c, classes, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), lp, act,
                                           fs, fb, red.data_ptr(), chan.data_ptr(), st))
        dx = torch.full((n, c), 7.0, device=device)
        check(lib.cp_bn_act_bwd_apply_f32(xd.data_ptr(), c, dyd.data_ptr(), c, n, c, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), lp, act,
                                          fs, fb, chan.data_ptr(), float(n), None, dx.data_ptr(), c, 0, st))
        r = red.cpu().numpy().reshape(classes, c, 2)
        assert rel(r[..., 0], bt.grad.numpy()) < 2e-5, "d beta"
        assert rel(r[..., 1], gt.grad.numpy()) < 2e-5, "d gamma"
        assert rel(dx.cpu().numpy(), xt.grad.numpy()) < 5e-5, "dx"
        # row scale + accumulate
        check(lib.cp_bn_act_bwd_apply_f32(xd.data_ptr(), c, dyd.data_ptr(), c, n, c, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), lp, act,
                                          fs, fb, chan.data_ptr(), float(n), d(rs).data_ptr(), dx.data_ptr(), c, 1, st))
        assert rel(dx.cpu().numpy(), xt.grad.numpy() * (1 + rs[:, None])) < 5e-5
    if act:
        # with the forward's tables the branch is the forward's, bit for bit: at an element whose pre-activation is a rounding error away
        # from zero the gradient follows the sign of the forward's OUTPUT.  Build such elements: x chosen so that fma(x, scale, shift) is
        # the smallest positive / negative float the forward can produce at that element, dy = 1.
        y0 = y.cpu().numpy()
        one = torch.ones(n, c, device=device)
        check(lib.cp_bn_act_bwd_reduce_f32(xd.data_ptr(), c, one.data_ptr(), c, n, c, classes, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), lp, act,
                                           scd.data_ptr(), shd.data_ptr(), red.data_ptr(), chan.data_ptr(), st))
        slope = np.where(y0 > 0, 1.0, float(np.float32(0.1)) if act == 2 else 0.0) * (y0 != 0 if act == 2 else 1)
        want = np.zeros((classes, c))
        np.add.at(want, lab.astype(np.int64), slope)
        assert np.allclose(red.cpu().numpy().reshape(classes, c, 2)[..., 0], want, rtol=0, atol=1e-9), "d beta with dy = 1 counts the forward's own branches"


@pytest.mark.parametrize("classes,act,cout,off", [(1, 2, 9, 0), (9, 2, 18, 32), (5, 1, 27, 32), (14, 2, 14, 0)])
def test_fused_head_normalisation_kernels(device, hip_lib, classes, act, cout, off):
    """cp_head1x1_{fwd,wgrad}_affine_f32 and cp_head1x1_bn_bwd_{reduce,apply}_f32 (blocks 5 / 10 -> head without a stored activation or a
    stored head data gradient) against the separate passes they replace, and against fp64 autograd of normalise -> activate -> 1x1 head."""
    lib = hip_lib
    from casapose_amd._lib import check

    rng = np.random.default_rng(classes * 100 + cout)
    b, h, w, c, ldo = 2, 16, 24, 32, 64
    n = b * h * w
    x = (rng.standard_normal((n, c)) * 2 + 0.5).astype(np.float32)
    lab = blob_labels(b, h, w, classes, 3).reshape(n) if classes > 1 else np.zeros(n, np.uint8)
    gamma = (1 + 0.3 * rng.standard_normal((classes, c))).astype(np.float32)
    beta = (0.2 * rng.standard_normal((classes, c))).astype(np.float32)
    wk = (0.2 * rng.standard_normal((c, cout))).astype(np.float32)
    dout = rng.standard_normal((n, ldo)).astype(np.float32)   # the other head's gradient / padding sits beside this head's columns
    rs = rng.uniform(0.5, 2.0, n).astype(np.float32)
    st = torch.cuda.current_stream(device).cuda_stream
    keep = []

    def d(a):
        keep.append(torch.from_numpy(np.ascontiguousarray(a)).to(device))
        return keep[-1]

    xd, labd, wd, doutd, rsd = d(x), d(lab), d(wk), d(dout), d(rs)
    lp = labd.data_ptr() if classes > 1 else None
    x64 = x.astype(np.float64)
    meanv = x64.mean(0)
    varv = (x64 ** 2).mean(0) - meanv ** 2
    rstd = 1.0 / np.sqrt(varv + R.BN_EPS)
    scale = (gamma * rstd[None]).astype(np.float32)
    shift = (beta - gamma * (meanv * rstd)[None]).astype(np.float32)
    scd, shd, md, rd, gd, bd = d(scale), d(shift), d(meanv.astype(np.float32)), d(rstd.astype(np.float32)), d(gamma), d(beta)
    # ---- the separate passes -------------------------------------------------------------------------------------------
    y = torch.empty(n, c, device=device)
    check(lib.cp_affine_act_f32(xd.data_ptr(), n, c, c, scd.data_ptr(), shd.data_ptr(), lp, act, y.data_ptr(), c, st))
    out0 = torch.zeros(n, 40, device=device)
    check(lib.cp_head1x1_fwd_f32(y.data_ptr(), c, n, wd.data_ptr(), cout, out0.data_ptr(), 40, st))
    dptr = doutd.data_ptr() + 4 * off
    dw0 = torch.zeros(c, cout, device=device)
    check(lib.cp_head1x1_wgrad_f32(y.data_ptr(), c, dptr, ldo, n, cout, dw0.data_ptr(), 0, st))
    gy = torch.empty(n, c, device=device)
    check(lib.cp_head1x1_dgrad_f32(dptr, ldo, 32, n, wd.data_ptr(), cout, gy.data_ptr(), c, 0, st))
    red0 = torch.zeros(classes * c * 2, dtype=torch.float64, device=device)
    chan0 = torch.zeros(c * 2, dtype=torch.float64, device=device)
    check(lib.cp_bn_act_bwd_reduce_f32(xd.data_ptr(), c, gy.data_ptr(), c, n, c, classes, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), lp, act,
                                       scd.data_ptr(), shd.data_ptr(), red0.data_ptr(), chan0.data_ptr(), st))
    dx0 = torch.empty(n, c, device=device)
    check(lib.cp_bn_act_bwd_apply_f32(xd.data_ptr(), c, gy.data_ptr(), c, n, c, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), lp, act,
                                      scd.data_ptr(), shd.data_ptr(), chan0.data_ptr(), float(n), rsd.data_ptr(), dx0.data_ptr(), c, 0, st))
    # ---- fused ---------------------------------------------------------------------------------------------------------
    out1 = torch.zeros(n, 40, device=device)
    check(lib.cp_head1x1_fwd_affine_f32(xd.data_ptr(), c, n, scd.data_ptr(), shd.data_ptr(), lp, classes, act, wd.data_ptr(), cout, out1.data_ptr(), 40, st))
    assert torch.equal(out0, out1), "same fma, same activation, same MFMA chain: bit-identical head output"
    # ... and as the writer of COMPLETE records: [dense prefix rows | this head's columns], bit for bit, nothing written beyond them
    for pre_n, ld_pre in ((9, 9), (14, 16), (1, 3)):
        pref = torch.randn(n, ld_pre, device=device)
        ldr = pre_n + cout + (0 if pre_n == 9 else 3)
        rec = torch.full((n, ldr), 7.0, device=device)
        check(lib.cp_head1x1_fwd_affine_record_f32(xd.data_ptr(), c, n, scd.data_ptr(), shd.data_ptr(), lp, classes, act, wd.data_ptr(), cout, pref.data_ptr(),
                                                   ld_pre, pre_n, rec.data_ptr(), ldr, st))
        assert torch.equal(rec[:, :pre_n], pref[:, :pre_n]) and torch.equal(rec[:, pre_n:pre_n + cout], out1[:, :cout])
        assert bool((rec[:, pre_n + cout:] == 7.0).all())
    dw1 = torch.full((c, cout), 3.0, device=device)
    check(lib.cp_head1x1_wgrad_affine_f32(xd.data_ptr(), c, scd.data_ptr(), shd.data_ptr(), lp, classes, act, dptr, ldo, n, cout, dw1.data_ptr(), 1, st))
    assert rel(dw1.cpu().numpy() - 3.0, dw0.cpu().numpy()) < 1e-5
    red1 = torch.full((classes * c * 2,), 5.0, dtype=torch.float64, device=device)
    chan1 = torch.full((c * 2,), 5.0, dtype=torch.float64, device=device)
    args = (xd.data_ptr(), c, dptr, ldo, 32, n, wd.data_ptr(), cout, md.data_ptr(), rd.data_ptr(), gd.data_ptr(), scd.data_ptr(), shd.data_ptr(), lp, classes, act)
    check(lib.cp_head1x1_bn_bwd_reduce_f32(*args, red1.data_ptr(), chan1.data_ptr(), st))
    assert rel(red1.cpu().numpy(), red0.cpu().numpy()) < 1e-9 and rel(chan1.cpu().numpy(), chan0.cpu().numpy()) < 1e-9
    dx1 = torch.full((n, c), 7.0, device=device)
    check(lib.cp_head1x1_bn_bwd_apply_f32(*args, chan1.data_ptr(), float(n), rsd.data_ptr(), dx1.data_ptr(), c, st))
    assert rel(dx1.cpu().numpy(), dx0.cpu().numpy()) < 1e-6
    # ---- fp64 autograd of the three layers -------------------------------------------------------------------------------
    xt = torch.tensor(x64, requires_grad=True)
    gt, bt = torch.tensor(gamma.astype(np.float64), requires_grad=True), torch.tensor(beta.astype(np.float64), requires_grad=True)
    wt = torch.tensor(wk.astype(np.float64), requires_grad=True)
    mean = xt.mean(0)
    var = ((xt - mean) ** 2).mean(0)
    xh = (xt - mean) / torch.sqrt(var + R.BN_EPS)
    li = torch.from_numpy(lab.astype(np.int64))
    t = gt[li] * xh + bt[li]
    yr = F.relu(t) if act == 1 else R.leaky_pair(t)
    o = yr @ wt
    assert rel(out1.cpu().numpy()[:, :cout], o.detach().numpy()) < 2e-5
    o.backward(torch.from_numpy(dout[:, off:off + cout].astype(np.float64)))
    assert rel(dw0.cpu().numpy(), wt.grad.numpy()) < 3e-5 and rel(dw1.cpu().numpy() - 3.0, wt.grad.numpy()) < 3e-5
    r = red1.cpu().numpy().reshape(classes, c, 2)
    assert rel(r[..., 0], bt.grad.numpy()) < 3e-5 and rel(r[..., 1], gt.grad.numpy()) < 3e-5
    check(lib.cp_head1x1_bn_bwd_apply_f32(*args, chan1.data_ptr(), float(n), None, dx1.data_ptr(), c, st))
    assert rel(dx1.cpu().numpy(), xt.grad.numpy()) < 5e-5


@pytest.mark.parametrize("c,cout,dil,hw", [(256, 256, 2, (15, 20)), (128, 384, 1, (12, 16)), (256, 128, 4, (9, 7))])
def test_winograd_transforms_with_fused_normalisation(device, hip_lib, c, cout, dil, hw):
    """cp_wino_input_transform_pre_f32 == cp_affine_act_f32 followed by the plain transform (bit for bit), and
    cp_wino_output_transform_stats_f32's table == cp_bn_stats_f32 of the raw output it writes (ragged tiles, dilation sub-grids, residual)."""
    lib = hip_lib
    from casapose_amd._lib import check

    rng = np.random.default_rng(c + cout)
    b, (h, w) = 2, hw
    n = b * h * w
    st = torch.cuda.current_stream(device).cuda_stream
    t, tp = C.c_int(), C.c_int()
    check(lib.cp_wino_tiles(b, h, w, dil, C.byref(t), C.byref(tp)))
    tp = tp.value
    x = torch.from_numpy((rng.standard_normal((n, c)) * 2 + 0.3).astype(np.float32)).to(device)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(device)
    sh = torch.from_numpy(rng.standard_normal(c).astype(np.float32)).to(device)
    for act in (1, 2, 0):
        y = torch.empty(n, c, device=device)
        check(lib.cp_affine_act_f32(x.data_ptr(), n, c, c, sc.data_ptr(), sh.data_ptr(), None, act, y.data_ptr(), c, st))
        v0 = torch.zeros(36 * tp * c, device=device)
        v1 = torch.zeros(36 * tp * c, device=device)
        check(lib.cp_wino_input_transform_f32(y.data_ptr(), c, c, b, h, w, dil, v0.data_ptr(), c, 0, st))
        check(lib.cp_wino_input_transform_pre_f32(x.data_ptr(), c, c, b, h, w, dil, v1.data_ptr(), c, 0, sc.data_ptr(), sh.data_ptr(), act, st))
        assert torch.equal(v0, v1), "act %d" % act
    # a per-channel factor alone (no shift table, no activation): the transform of the scaled tensor -- to rounding: the two instantiations contract
    # B^T d B's multiply-adds differently, every one of them a correct rounding of the same sum
    p2 = torch.from_numpy((2.0 ** rng.integers(-12, 13, c)).astype(np.float32)).to(device)
    zero = torch.zeros(c, device=device)
    check(lib.cp_wino_input_transform_pre_f32(x.data_ptr(), c, c, b, h, w, dil, v0.data_ptr(), c, 0, p2.data_ptr(), zero.data_ptr(), 0, st))
    check(lib.cp_wino_input_transform_pre_f32(x.data_ptr(), c, c, b, h, w, dil, v1.data_ptr(), c, 0, p2.data_ptr(), None, 0, st))
    v0c, v1c = v0.view(36 * tp, c), v1.view(36 * tp, c)
    assert float(((v0c - v1c).abs().amax(0) / v0c.abs().amax(0).clamp_min(1e-30)).max()) < 2e-6
    assert lib.cp_wino_input_transform_pre_f32(x.data_ptr(), c, c, b, h, w, dil, v1.data_ptr(), c, 0, p2.data_ptr(), None, 1, st) != 0   # an activation needs its shift
    m = torch.from_numpy(rng.standard_normal(36 * tp * cout).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(device)
    for r in (None, res):
        raw0 = torch.zeros(n, cout, device=device)
        raw1 = torch.zeros(n, cout, device=device)
        stats = torch.full((2 * cout,), 9.0, dtype=torch.float64, device=device)
        rp = r.data_ptr() if r is not None else None
        check(lib.cp_wino_output_transform_f32(m.data_ptr(), cout, b, h, w, dil, rp, cout, None, None, None, 0, raw0.data_ptr(), cout, None, cout, st))
        check(lib.cp_wino_output_transform_stats_f32(m.data_ptr(), cout, b, h, w, dil, rp, cout, None, None, None, 0, raw1.data_ptr(), cout, None, cout,
                                                     stats.data_ptr(), st))
        assert torch.equal(raw0, raw1)
        want = torch.zeros(2 * cout, dtype=torch.float64, device=device)
        check(lib.cp_bn_stats_f32(raw0.data_ptr(), n, cout, cout, want.data_ptr(), st))
        r64 = raw0.cpu().numpy().astype(np.float64)
        assert rel(want.cpu().numpy(), np.concatenate([r64.sum(0), (r64 ** 2).sum(0)])) < 1e-12
        # fp64 from the first addition on both routes: only the summation order differs
        assert np.abs(stats.cpu().numpy() - want.cpu().numpy()).max() < 1e-12 * np.abs(r64).max() ** 2 * n


def test_fused_normalisation_plan_equals_the_separate_passes(device, monkeypatch):
    """TrainPlan with the fused normalisation (default) against CASAPOSE_FUSE_NORM=0: same outputs, same losses, same gradient."""
    from casapose_amd.train_engine import BnActOp

    b, h, w, k = 2, 64, 64, 5
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("CASAPOSE_FUSE_NORM", fuse)
        params, store, plan, img, lab, kpts = _setup(device, b, h, w, k)
        fused = [op.name for op in plan.ops if isinstance(op, BnActOp) and (op.head is not None or op.stats_from is not None or op.consumer is not None)]
        assert bool(fused) == (fuse == "1"), fused
        if fuse == "1":
            assert {"pv_block_5_bn", "pv_block_10_clade"} <= set(fused), fused
        stream = torch.cuda.current_stream(device).cuda_stream
        plan.refresh_weights(stream)
        labd = torch.from_numpy(lab).to(device)
        out = plan.forward(torch.from_numpy(img).to(device), cond_labels=labd).clone()
        sums = plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(device), 1.0, 0.5, 0.015, filter_with_segmentation=False).cpu().numpy().copy()
        plan.backward()
        torch.cuda.synchronize()
        res[fuse] = (out.cpu().numpy(), sums, store.grad.cpu().numpy().copy(), {n: v.cpu().numpy().copy() for n, v in store.state.items()})
    assert rel(res["1"][0], res["0"][0]) < 1e-5
    assert np.allclose(res["1"][1], res["0"][1], rtol=1e-5)
    assert rel_l2(res["1"][2], res["0"][2]) < 1e-4
    for n in res["0"][3]:   # moving statistics
        assert rel(res["1"][3][n], res["0"][3][n]) < 1e-5, n


def test_resampling_adjoints(device, hip_lib):
    lib = hip_lib
    from casapose_amd._lib import check

    rng = np.random.default_rng(0)
    b, h, w, c = 2, 10, 14, 32
    st = torch.cuda.current_stream(device).cuda_stream
    keep = []

    def d(a):
        keep.append(torch.from_numpy(np.ascontiguousarray(a)).to(device))
        return keep[-1]

    # max pool (post-ReLU input, like the reference graph)
    x = np.maximum(rng.standard_normal((b, h, w, c)), 0).astype(np.float32)
    xt = torch.tensor(x.astype(np.float64), requires_grad=True)
    yr = R.maxpool_zero_pad(xt)
    dy = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    (yr * torch.from_numpy(dy.astype(np.float64))).sum().backward()
    dx = torch.empty(b, h, w, c, device=device)
    check(lib.cp_maxpool3x3s2_bwd_f32(d(x).data_ptr(), d(dy).data_ptr(), b, h, w, c, dx.data_ptr(), 0, st))
    got = dx.cpu().numpy()
    pos = x > 0  # at exact zeros the following ReLU blocks the gradient; ties there are immaterial
    assert rel(got[pos], xt.grad.numpy()[pos]) < 1e-6
    # the training plan's pair: forward recording the arg-max tap, adjoint routed by it (no x reads) -- same output, same dx, odd sizes too
    for hh, ww in ((h, w), (9, 13)):
        xo = np.maximum(rng.standard_normal((b, hh, ww, c)), 0).astype(np.float32)
        ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        dyo = rng.standard_normal((b, ho, wo, c)).astype(np.float32)
        y0 = torch.empty(b, ho, wo, c, device=device)
        y1 = torch.empty(b, ho, wo, c, device=device)
        idx = torch.empty(b, ho, wo, c, dtype=torch.uint8, device=device)
        check(lib.cp_maxpool3x3s2_f32(d(xo).data_ptr(), b, hh, ww, c, None, None, 0, y0.data_ptr(), st))
        check(lib.cp_maxpool3x3s2_idx_f32(d(xo).data_ptr(), b, hh, ww, c, y1.data_ptr(), idx.data_ptr(), st))
        assert torch.equal(y0, y1) and int(idx.max()) <= 8
        dx0 = torch.full((b, hh, ww, c), 2.0, device=device)
        dx1 = torch.full((b, hh, ww, c), 2.0, device=device)
        for acc in (0, 1):
            check(lib.cp_maxpool3x3s2_bwd_f32(d(xo).data_ptr(), d(dyo).data_ptr(), b, hh, ww, c, dx0.data_ptr(), acc, st))
            check(lib.cp_maxpool3x3s2_bwd_idx_f32(idx.data_ptr(), d(dyo).data_ptr(), b, hh, ww, c, dx1.data_ptr(), acc, st))
            assert torch.equal(dx0, dx1)
    # bilinear x2
    x = rng.standard_normal((b, h, w, c)).astype(np.float32)
    xt = torch.tensor(x.astype(np.float64), requires_grad=True)
    yr = R.bilinear_x2(xt)
    dy = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    (yr * torch.from_numpy(dy.astype(np.float64))).sum().backward()
    check(lib.cp_upsample_bilinear_x2_bwd_f32(d(dy).data_ptr(), c, b, h, w, c, dx.data_ptr(), st))
    assert rel(dx.cpu().numpy(), xt.grad.numpy()) < 1e-5
    # guided nearest x2
    lab_hi = blob_labels(b, 2 * h, 2 * w, 4, 3)
    lab_lo = lab_hi[:, ::2, ::2]
    sel = R.guided_select(torch.from_numpy(lab_lo.astype(np.int64)), torch.from_numpy(lab_hi.astype(np.int64))).numpy().astype(np.uint8)
    xt = torch.tensor(x.astype(np.float64), requires_grad=True)
    yr = R.guided_upsample(xt, torch.from_numpy(lab_lo.astype(np.int64)), torch.from_numpy(lab_hi.astype(np.int64)))
    (yr * torch.from_numpy(dy.astype(np.float64))).sum().backward()
    check(lib.cp_guided_upsample_x2_bwd_f32(d(dy).data_ptr(), c, d(sel).data_ptr(), b, h, w, c, dx.data_ptr(), st))
    assert rel(dx.cpu().numpy(), xt.grad.numpy()) < 1e-5


@pytest.mark.parametrize("filt,high", [(True, False), (False, False), (True, True)])
def test_pose_loss_value_and_gradient(device, hip_lib, filt, high):
    lib = hip_lib
    from casapose_amd._lib import check

    rng = np.random.default_rng(11)
    b, h, w, k, kp = 2, 24, 32, 5, 9
    ld = k + 3 * kp
    out = rng.standard_normal((b, h, w, ld)).astype(np.float32)
    lab = blob_labels(b, h, w, k, 2)
    out[..., :k] += 3.0 * np.eye(k, dtype=np.float32)[lab] * (rng.uniform(size=(b, h, w, 1)) < 0.7)  # mostly-correct prediction
    out[..., k:k + 2 * kp] *= np.where(rng.uniform(size=(b, h, w, 1)) < 0.5, 0.3, 3.0)  # both smooth-L1 branches
    kpts = rng.uniform(-10, 40, (b, k - 1, kp, 2)).astype(np.float32)
    ot = torch.tensor(out.astype(np.float64), requires_grad=True)
    if high:  # make two objects of image 0 point exactly at their keypoints so that they survive the proxy-error filter, the rest fails it
        yy, xx = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
        for o in (1, 2):
            m = lab[0] == o
            d = kpts[0, o - 1][None, None] - np.stack([yy, xx], -1)[:, :, None, :]
            d = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
            an = rng.normal(0, 0.05, d.shape[:-1])  # a little angular noise and a non-unit length: small but non-zero losses
            d = np.stack([np.cos(an) * d[..., 0] - np.sin(an) * d[..., 1], np.sin(an) * d[..., 0] + np.cos(an) * d[..., 1]], -1) * 1.3
            out[0][m, k:k + 2 * kp] = d.reshape(h, w, -1)[m]
            out[0][m, :k] = 5.0 * np.eye(k, dtype=np.float32)[o]
    ot = torch.tensor(out.astype(np.float64), requires_grad=True)
    ml, vl, pl = R.losses(ot, torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64)), k, kp, filt, high)
    wts = (1.0, 0.5, 0.015)
    (wts[0] * ml + wts[1] * vl + wts[2] * pl).backward()
    st = torch.cuda.current_stream(device).cuda_stream
    d = lambda a: torch.from_numpy(a).to(device)
    od, labd, kd = d(out), d(lab), d(kpts)
    ws = torch.empty(lib.cp_pose_loss_workspace_bytes(b, h, w), dtype=torch.uint8, device=device)
    dout = torch.full((b, h, w, 64), 5.0, device=device)
    sums = torch.zeros(3, dtype=torch.float64, device=device)
    vals = torch.zeros(b, k - 1, device=device)
    check(lib.cp_pose_loss_f32(od.data_ptr(), ld, k, kp, labd.data_ptr(), labd.data_ptr(), kd.data_ptr(), k - 1, b, h, w, int(filt), int(high), *wts, ws.data_ptr(),
                               dout.data_ptr(), 64, 32, sums.data_ptr(), vals.data_ptr(), st))
    s = sums.cpu().numpy()
    if high:  # per-object proxy values (proxy_voting_dist): the two exact objects pass, at least one noisy object is filtered out
        v = vals.cpu().numpy()
        assert (v[0, :2] < 5).all() and (v >= 5).any()
    assert abs(s[0] - ml.item()) < 1e-5 * abs(ml.item())
    assert abs(s[1] - vl.item()) < 1e-5 * abs(vl.item())
    assert abs(s[2] - pl.item()) < 1e-5 * abs(pl.item())
    g = dout.cpu().numpy()
    gr = ot.grad.numpy()
    assert rel(g[..., :k], gr[..., :k]) < 2e-5
    assert rel(g[..., 32:32 + 2 * kp], gr[..., k:k + 2 * kp]) < 2e-5
    assert np.all(g[..., k:32] == 0) and np.all(g[..., 32 + 2 * kp:] == 0)


def test_adam_matches_keras_formula(device, hip_lib):
    lib = hip_lib
    from casapose_amd._lib import check

    rng = np.random.default_rng(5)
    n = 10007
    p = rng.standard_normal(n).astype(np.float32)
    m, v = np.zeros(n), np.zeros(n)
    pr = p.astype(np.float64)
    pd, md, vd = torch.from_numpy(p).to(device), torch.zeros(n, device=device), torch.zeros(n, device=device)
    st = torch.cuda.current_stream(device).cuda_stream
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        gd = torch.from_numpy(g).to(device)
        check(lib.cp_adam_step_f32(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, lr, b1, b2, eps, step, 1.0, st))
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g.astype(np.float64) ** 2
        pr -= lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step) * m / (np.sqrt(v) + eps)
    assert rel(pd.cpu().numpy(), pr) < 1e-6


# --------------------------------------------------------------------------------------------------
# whole network
# --------------------------------------------------------------------------------------------------
def _setup(device, b, h, w, k, seed=1237, partial=(True,) * 5, guided=(False, True, True, True, False), bilinear=(False,) * 5, sharing=None):
    from casapose_amd.train_engine import ParamStore, TrainPlan

    v = 27
    sharing = sharing or {}
    params = O.init_params(k, v, seed=seed, dtype=np.float32, partial=partial, **sharing)
    rng = np.random.default_rng(seed)
    for name in params:  # non-trivial normalisation parameters
        if name.endswith(".gamma"):
            params[name] = (1 + 0.2 * rng.standard_normal(params[name].shape)).astype(np.float32)
        if name.endswith(".beta"):
            params[name] = (0.1 * rng.standard_normal(params[name].shape)).astype(np.float32)
    store = ParamStore(params, device)
    plan = TrainPlan(store, k, v, b, h, w, partial=partial, guided=guided, bilinear=bilinear, **sharing)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    lab = blob_labels(b, h, w, k, seed + 1)
    kpts = rng.uniform(0, min(h, w), (b, k - 1, 9, 2)).astype(np.float32)
    return params, store, plan, img, lab, kpts


@pytest.mark.parametrize("variant", ["casapose_c_gcu5", "casapose_c_gcu3", "casapose_c", "casapose_c_gcu4_bilat",
                                     "casapose_c_gcu5_sw5", "casapose_c_gcu4_sw1", "casapose_c_gcu5_sw1", "casapose_c_gcu4_sw2"])
def test_train_forward_backward_matches_autograd(device, variant):
    b, h, w, k = 2, 32, 48, 4
    part, guid = O.VARIANTS[variant]
    bil = O.BILINEAR_GUIDED.get(variant, (False,) * 5)
    sharing = O.SHARED.get(variant, {})
    params, store, plan, img, lab, kpts = _setup(device, b, h, w, k, partial=part, guided=guid, bilinear=bil, sharing=sharing)
    stream = torch.cuda.current_stream(device).cuda_stream
    plan.refresh_weights(stream)
    labd = torch.from_numpy(lab).to(device)
    out = plan.forward(torch.from_numpy(img).to(device), cond_labels=labd)
    p64 = R.to_torch(params)
    stats, pre = {}, {}
    ref = R.forward_train(p64, torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), stats, partial=part, guided=guid, bilinear=bil,
                          preact_out=pre, **sharing)
    got = out.cpu().numpy()
    assert rel(got[..., :k], ref.detach().numpy()[..., :k]) < 1e-3
    assert rel(got[..., k:], ref.detach().numpy()[..., k:]) < 1e-3
    wts = (1.0, 0.5, 0.015)
    sums = plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(device), *wts, filter_with_segmentation=False)
    ml, vl, pl = R.losses(ref, torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64)), k, 9, False)
    s = sums.cpu().numpy()
    assert abs(s[0] - ml.item()) < 1e-3 * abs(ml.item()) and abs(s[1] - vl.item()) < 1e-3 * abs(vl.item()) and abs(s[2] - pl.item()) < 1e-3 * abs(pl.item())
    plan.backward()
    torch.cuda.synchronize()
    # ReLU / leaky kinks: the fp32 forward takes the other branch at the few elements whose fp64 pre-activation is a rounding error away
    # from zero (one such element moves a [classes, C] table reduced over 32 pixels by ~1e-2).  The device differentiates the function it
    # evaluated, so the reference is the fp64 gradient ON THE DEVICE'S BRANCHES; that those branches differ from the oracle's own only at
    # |pre-activation| < 1e-4 is asserted, and with it every variable is gated at 1e-3 relative L2 (round 2: 2e-2, no kink handling).
    pattern = plan.activation_pattern()
    flips, total, margin = R.kink_report(pattern, pre)
    assert margin < 1e-4 and flips < 1e-4 * total, (flips, total, margin)
    if flips:
        p64 = R.to_torch(params)
        ref = R.forward_train(p64, torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), None, partial=part, guided=guid,
                              bilinear=bil, act_pattern=pattern, **sharing)
        ml, vl, pl = R.losses(ref, torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64)), k, 9, False)
    (wts[0] * ml + wts[1] * vl + wts[2] * pl).backward()
    worst = {name: rel_l2(store.grad_view(name).cpu().numpy(), p64[name].grad.numpy()) for name in store.offsets}
    top = sorted(worst.items(), key=lambda t: -t[1])
    print("%s: %d kink flips (margin %.1e); worst gradient %s %.2e, median %.2e" % (variant, flips, margin, top[0][0], top[0][1], np.median(list(worst.values()))))
    bad = {n: e for n, e in worst.items() if e > 1e-3}
    assert not bad, "gradient mismatch (relative L2): %s" % sorted(bad.items(), key=lambda t: -t[1])[:10]
    # moving statistics were updated with the batch statistics
    mm = store.state["bn0.moving_mean"].cpu().numpy()
    exp = 0.99 * params["bn0.moving_mean"] + 0.01 * stats["bn0"][0].numpy()
    assert rel(mm, exp) < 1e-4


def test_train_forward_backward_bf16_conv_mode_within_its_gates(device, monkeypatch):
    """CASAPOSE_CONV_MODE=bf16 (BASELINE.json configs[2] "bf16 convs"): the 3x3 layers off the Winograd path with bf16 operands in forward,
    data gradient and weight gradient, the Winograd GEMMs with hi + mid planes.  SURVEY 8d gate for the bf16 path: outputs <= 3e-2 of the
    fp64 reference's range.  The GRADIENTS of this loss are not smooth in the outputs (smooth-L1 / proxy-voting kinks, arg-max conditioning):
    an output error of 1-2 % of range turns into ~20 % relative L2 on every variable (measured identically with the Winograd GEMMs in fp32,
    `tools/debug/bf16_grad_probe.py`), so they are held to a sanity gate only -- right direction, right size -- while each bf16 kernel has its
    own 2e-2 / 3e-2 parity test (tests/test_gpu_hsplit.py, tests/test_gpu_wgrad_split.py, tests/test_gpu_conv.py)."""
    monkeypatch.setenv("CASAPOSE_CONV_MODE", "bf16")
    b, h, w, k = 2, 64, 96, 4
    part, guid = O.VARIANTS["casapose_c_gcu5"]
    params, store, plan, img, lab, kpts = _setup(device, b, h, w, k, partial=part, guided=guid, bilinear=(False,) * 5, sharing={})
    assert any(getattr(op, "wgrad_planes", lambda: 0)() == 1 for op in plan.ops), "the bf16 weight-gradient kernel is not on the path"
    stream = torch.cuda.current_stream(device).cuda_stream
    plan.refresh_weights(stream)
    labd = torch.from_numpy(lab).to(device)
    out = plan.forward(torch.from_numpy(img).to(device), cond_labels=labd)
    p64 = R.to_torch(params)
    pre = {}
    ref = R.forward_train(p64, torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), {}, partial=part, guided=guid,
                          bilinear=(False,) * 5, preact_out=pre)
    got = out.cpu().numpy()
    e_seg, e_vec = rel(got[..., :k], ref.detach().numpy()[..., :k]), rel(got[..., k:], ref.detach().numpy()[..., k:])
    assert e_seg < 3e-2 and e_vec < 3e-2, (e_seg, e_vec)
    assert max(e_seg, e_vec) > 1e-5, "these are bf16 operands: an fp32-like error means the mode did not take effect"
    wts = (1.0, 0.5, 0.015)
    plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(device), *wts, filter_with_segmentation=False)
    ml, vl, pl = R.losses(ref, torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64)), k, 9, False)
    (wts[0] * ml + wts[1] * vl + wts[2] * pl).backward()
    plan.backward()
    torch.cuda.synchronize()
    worst = {name: rel_l2(store.grad_view(name).cpu().numpy(), p64[name].grad.numpy()) for name in store.offsets}
    bad = {n: e for n, e in worst.items() if e > 0.5}
    assert not bad, "gradient mismatch (relative L2): %s" % sorted(bad.items(), key=lambda t: -t[1])[:10]
    assert np.median(list(worst.values())) < 0.3, np.median(list(worst.values()))
    for name in ("conv0.kernel", "stage4_unit2_conv2.kernel", "pv_block_10_prepare_conv2d.weights"):
        g, gr = store.grad_view(name).cpu().numpy().ravel().astype(np.float64), p64[name].grad.numpy().ravel()
        assert g @ gr / (np.linalg.norm(g) * np.linalg.norm(gr)) > 0.9, name
    # how much of that is branch flips (1-2 % output error changes the sign of thousands of ReLU / leaky pre-activations) and how much is the
    # rounding of the bf16 products themselves: the same comparison against the fp64 oracle on the DEVICE's branches.  Measured: the median
    # drops from ~0.2 to ~0.08, the most cancellation-prone variables (decoder-2 partial-convolution weights) stay at 0.4-0.5 -- the same
    # ~60x amplification of the operand rounding (2^-9 here, 2^-24 in the fp32-equivalent modes, where these variables sit at 2e-5 against a
    # median of 4e-6).  Whether such gradients TRAIN is what test_bf16_conv_mode_converges_like_the_fp32_equivalent_mode checks.
    pattern = plan.activation_pattern()
    flips, total, margin = R.kink_report(pattern, pre)
    p64b = R.to_torch(params)
    refb = R.forward_train(p64b, torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), {}, partial=part, guided=guid,
                           bilinear=(False,) * 5, act_pattern=pattern)
    ml, vl, pl = R.losses(refb, torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64)), k, 9, False)
    (wts[0] * ml + wts[1] * vl + wts[2] * pl).backward()
    worst_b = {name: rel_l2(store.grad_view(name).cpu().numpy(), p64b[name].grad.numpy()) for name in store.offsets}
    print("bf16 mode: %d of %d branches flipped (largest |pre-activation| %.2e); gradient error median %.3f / worst %.3f against the oracle's own "
          "branches, median %.3f / worst %.3f on the device's branches" % (flips, total, margin, np.median(list(worst.values())), max(worst.values()),
                                                                         np.median(list(worst_b.values())), max(worst_b.values())))
    assert np.median(list(worst_b.values())) < 0.15 and max(worst_b.values()) < 0.7, sorted(worst_b.items(), key=lambda t: -t[1])[:5]
    assert np.median(list(worst_b.values())) < np.median(list(worst.values()))


def test_train_steps_reduce_the_loss(device):
    b, h, w, k = 2, 32, 32, 3
    params, store, plan, img, lab, kpts = _setup(device, b, h, w, k, seed=99)
    stream = torch.cuda.current_stream(device).cuda_stream
    plan.refresh_weights(stream)
    imgd, labd, kd = torch.from_numpy(img).to(device), torch.from_numpy(lab).to(device), torch.from_numpy(kpts).to(device)
    hist = []
    for it in range(12):
        s = plan.train_step(imgd, labd, labd, kd, lr=1e-3, cond_labels=labd, weights=(1.0, 0.5, 0.015))
        hist.append(s.cpu().numpy().copy())
    hist = np.array(hist)
    total = hist[:, 0] + 0.5 * hist[:, 1] + 0.015 * hist[:, 2]
    assert np.all(np.isfinite(hist))
    assert total[-1] < 0.8 * total[0], "loss did not go down: %s" % total


def test_bf16_conv_mode_converges_like_the_fp32_equivalent_mode(device, monkeypatch):
    """BASELINE.json configs[2] ("bf16 convs"): 30 optimisation steps from the same initialisation on the same batch, once with the training
    plan's default (exact three-way bf16 splits, fp32-equivalent) and once with CASAPOSE_CONV_MODE=bf16 (operands of the 3x3 layers rounded
    to bf16 in forward, data gradient and weight gradient; Winograd GEMMs on hi + mid planes).  The bf16 run must train: its loss curve stays
    within 3 % (+ 0.01 absolute) of the fp32-equivalent curve at EVERY step (measured: <= 0.4 %) and ends below 60 % of the initial loss."""
    b, h, w, k, steps = 4, 64, 64, 4, 30
    curves = {}
    for mode in ("split", "bf16"):
        monkeypatch.setenv("CASAPOSE_CONV_MODE", mode)
        params, store, plan, img, lab, kpts = _setup(device, b, h, w, k, seed=77)
        if mode == "bf16":
            assert any(getattr(op, "wgrad_planes", lambda: 0)() == 1 for op in plan.ops), "the bf16 kernels are not on the path"
        stream = torch.cuda.current_stream(device).cuda_stream
        plan.refresh_weights(stream)
        imgd, labd, kd = torch.from_numpy(img).to(device), torch.from_numpy(lab).to(device), torch.from_numpy(kpts).to(device)
        hist = []
        for _ in range(steps):
            s = plan.train_step(imgd, labd, labd, kd, lr=1e-3, cond_labels=labd, weights=(1.0, 0.5, 0.015)).cpu().numpy()
            hist.append(s[0] + 0.5 * s[1] + 0.015 * s[2])
        curves[mode] = np.array(hist)
    a, c = curves["split"], curves["bf16"]
    print("loss, exact-split mode: %s" % np.round(a[::3], 4))
    print("loss, bf16 mode       : %s" % np.round(c[::3], 4))
    assert np.all(np.isfinite(c)) and c[-1] < 0.6 * c[0] and a[-1] < 0.6 * a[0], (a[0], a[-1], c[0], c[-1])
    assert np.all(np.abs(c - a) <= 0.03 * a + 0.01), "bf16 curve leaves the band: max deviation %.3g at step %d" % (
        np.abs(c - a).max(), int(np.abs(c - a).argmax()))
    assert np.abs(c - a).max() > 1e-7, "identical curves: the bf16 mode did not take effect"


def test_fp16_pair_training_tracks_the_exact_split_training(device, monkeypatch):
    """Round 6: the plan's default runs forward AND backward GEMMs on fp16 pairs (three products per fp32 product, gradients behind powers of two that
    follow device-side maxima).  30 optimisation steps from the same initialisation on the same batch, once in that default and once with the exact
    three-way bf16 split in both directions: the loss curves must agree to 1 % (+ 0.005 absolute) at EVERY step -- the band the bf16-operand mode gets
    3 % of -- and both must train.  With a reading of the range slots every 4 steps, so that exponents are re-judged several times on the way."""
    from casapose_amd import engine as E
    from casapose_amd import train_engine as TE

    if not E.TRAIN_WINO_GEMM_SPLIT:
        pytest.skip("CASAPOSE_WINO_GEMM=f32: no fp16-pair GEMMs in this process")
    monkeypatch.setattr(TE, "F16X2_TRAIN_CHECK_EVERY", 4)
    monkeypatch.delenv("CASAPOSE_CONV_MODE", raising=False)
    b, h, w, k, steps = 4, 64, 64, 4, 30
    curves, plans = {}, {}
    for mode in ("split", "f16x2"):
        monkeypatch.setenv("CASAPOSE_TRAIN_FWD", mode)
        monkeypatch.setenv("CASAPOSE_TRAIN_BWD", mode)
        params, store, plan, img, lab, kpts = _setup(device, b, h, w, k, seed=77)
        stream = torch.cuda.current_stream(device).cuda_stream
        plan.refresh_weights(stream)
        imgd, labd, kd = torch.from_numpy(img).to(device), torch.from_numpy(lab).to(device), torch.from_numpy(kpts).to(device)
        hist = []
        for _ in range(steps):
            s = plan.train_step(imgd, labd, labd, kd, lr=1e-3, cond_labels=labd, weights=(1.0, 0.5, 0.015)).cpu().numpy()
            hist.append(s[0] + 0.5 * s[1] + 0.015 * s[2])
        curves[mode], plans[mode] = np.array(hist), plan
    a, c = curves["split"], curves["f16x2"]
    print("loss, exact split: %s" % np.round(a[::3], 4))
    print("loss, fp16 pairs : %s" % np.round(c[::3], 4))
    slots = plans["f16x2"]._bwd_slots()
    assert slots and not plans["split"]._bwd_slots()
    on = sum(1 for _, f, e in slots if (f["on"] if e == "direct" else f["e"] is not None))
    assert on >= len(slots) // 2 and plans["f16x2"].f16x2_checks >= 3, (on, len(slots), plans["f16x2"].f16x2_checks)
    assert np.all(np.isfinite(c)) and c[-1] < 0.6 * c[0] and a[-1] < 0.6 * a[0], (a[0], a[-1], c[0], c[-1])
    assert np.all(np.abs(c - a) <= 0.01 * a + 0.005), "fp16-pair curve leaves the band: max deviation %.3g at step %d" % (np.abs(c - a).max(), int(np.abs(c - a).argmax()))


# --------------------------------------------------------------------------------------------------
# keypoint reprojection loss through the LS voter
# --------------------------------------------------------------------------------------------------
def _kp_case(seed, b, h, w, k):
    rng = np.random.default_rng(seed)
    kp = 9
    lab = blob_labels(b, h, w, k, seed + 3)
    kpts = rng.uniform(0.1 * h, 0.9 * h, (b, k - 1, kp, 2))
    yy, xx = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
    dirs = np.zeros((b, h, w, kp, 2))
    for n in range(b):
        for o in range(1, k):
            m = lab[n] == o
            d = kpts[n, o - 1][None, None] - np.stack([yy, xx], -1)[:, :, None, :]
            d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
            dirs[n][m] = d[m]
    ang = rng.normal(0, 0.15, (b, h, w, kp))
    ca, sa = np.cos(ang), np.sin(ang)
    dirs = np.stack([ca * dirs[..., 0] - sa * dirs[..., 1], sa * dirs[..., 0] + ca * dirs[..., 1]], -1) * rng.uniform(0.5, 1.5, (b, h, w, kp, 1))
    conf = rng.normal(0, 1, (b, h, w, kp))
    logits = rng.normal(0, 1, (b, h, w, k)) + 4.0 * np.eye(k)[lab]
    out = np.concatenate([logits, dirs.reshape(b, h, w, 2 * kp), conf], -1).astype(np.float32)
    offsets = np.stack([rng.uniform(0, 30, b), rng.uniform(0, 60, b), np.zeros(b), np.zeros(b), rng.uniform(-5, 5, b), rng.uniform(-5, 5, b),
                        rng.uniform(-20, 20, b), rng.uniform(0.8, 1.2, b), np.full(b, 640.0), np.full(b, 480.0)], 1)
    A = R.crop_to_image_affine(offsets)
    # ground-truth image keypoints: the true crop keypoints mapped to the image, plus a few pixels of "pose error"
    xy = kpts[..., ::-1]
    gt = np.stack([A[:, None, None, 0, 0] * xy[..., 0] + A[:, None, None, 0, 1] * xy[..., 1] + A[:, None, None, 0, 2],
                   A[:, None, None, 1, 0] * xy[..., 0] + A[:, None, None, 1, 1] * xy[..., 1] + A[:, None, None, 1, 2]], -1)
    gt += rng.normal(0, 3.0, gt.shape)
    gt[0, 0] += 40.0  # one object beyond the soft cap
    return lab, out, offsets, A, gt


@pytest.mark.parametrize("conf_reg", [False, True])
@pytest.mark.parametrize("k", [5, 9])
def test_keypoint_loss_and_voter_backward(device, conf_reg, k, monkeypatch):
    from casapose_amd.train_engine import ParamStore, TrainPlan, crop_to_image_affine

    b, h, w, kp = 2, 64, 64, 9  # record length k + 27 = 32 floats, or the production record of 36 (k = 9: the specialised voter kernels)
    lab, out, offsets, A, gt = _kp_case(21, b, h, w, k)
    # reference
    ot = torch.tensor(out.astype(np.float64), requires_grad=True)
    labt = torch.from_numpy(lab.astype(np.int64))
    coords = R.ls_voting(labt, ot[..., k:k + 2 * kp], ot[..., k + 2 * kp:], k - 1)
    est = torch.argmax(ot[..., :k].detach(), -1)
    avail = torch.stack([((est == o).sum((1, 2)) > 50) & ((labt == o).sum((1, 2)) > 50) for o in range(1, k)], 1).double()
    loss = R.keypoint_reprojection_loss(coords, torch.from_numpy(gt), torch.from_numpy(A), avail, ot[..., k + 2 * kp:], labt, 12.5, conf_reg)
    kp_w = 0.007
    (kp_w * loss).backward()
    # device
    params = O.init_params(k, 27, seed=5, dtype=np.float32)
    plan = TrainPlan(ParamStore(params, device), k, 27, b, h, w)
    plan.out.copy_(torch.from_numpy(out))
    plan.dout.zero_()
    Ad = torch.from_numpy(crop_to_image_affine(offsets)).to(device)
    assert rel(Ad.cpu().numpy().reshape(b, 2, 3), A) < 1e-6
    val = plan.kp_loss_and_grad(torch.from_numpy(lab).to(device), torch.from_numpy(gt.astype(np.float32)).to(device), Ad, kp_w, max_pixel_error=12.5,
                                min_num=50, confidence_regularization=conf_reg, vote_with_gt=True)
    assert rel(plan.ls_coords.cpu().numpy(), coords.detach().numpy()) < 1e-4
    assert abs(val.item() - loss.item()) < 1e-4 * abs(loss.item())
    g = plan.dout.cpu().numpy()
    gr = ot.grad.numpy()
    assert rel(g[..., 32:32 + 2 * kp], gr[..., k:k + 2 * kp]) < 2e-3
    assert rel(g[..., 32 + 2 * kp:32 + 3 * kp], gr[..., k + 2 * kp:]) < 2e-3
    assert np.all(g[..., :32] == 0)
    if k == 9:   # the production-record kernels (16-byte accesses, compile-time channel positions) against the generic ones: same arithmetic
        monkeypatch.setenv("CP_LS_GENERIC", "1")
        plan.dout.zero_()
        plan.kp_loss_and_grad(torch.from_numpy(lab).to(device), torch.from_numpy(gt.astype(np.float32)).to(device), Ad, kp_w, max_pixel_error=12.5,
                              min_num=50, confidence_regularization=conf_reg, vote_with_gt=True)
        g2 = plan.dout.cpu().numpy()
        assert rel(g, g2) < 1e-5


@pytest.mark.parametrize("k,bpnp", [(5, False), (14, False), (5, True)])
def test_model_api_train_step(device, k, bpnp):
    """The reference's flow (train_casapose.py:494-611) through the factory model and casapose_amd.training.train_step;
    k = 14 is the 13-object configuration (config_13.ini) whose output record (41 floats) is not a multiple of 16 bytes."""
    from types import SimpleNamespace

    from casapose_amd.pose_models.tfkeras import Classifiers
    from casapose_amd.training import Adam, train_step
    from casapose_amd.utils.learning_rate_schedules import LossWeightHandler, PiecewiseConstantDecay

    b, h, w, kp = 2, 64, 64, 9
    lab, out, offsets, A, gt = _kp_case(7, b, h, w, k)
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=k, input_shape=(h, w, 3), input_segmentation_shape=(h, w, k), weights=None,
                                             base_model="resnet18", device=device, seed=3)
    rng = np.random.default_rng(3)
    img = rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)
    seg = np.eye(k, dtype=np.float32)[lab]
    before = net([img, seg], training=False).cpu().numpy()
    # keypoints: 3-D points on a plane in front of an identity camera so that the projection is known
    cam = np.array([[100.0, 0, 32.0], [0, 100.0, 32.0], [0, 0, 1]])
    p3d = rng.uniform(-20, 20, (b, k - 1, 1, kp, 3))
    poses = np.zeros((b, k - 1, 1, 3, 4))
    poses[..., :3, :3] = np.eye(3)
    poses[..., 2, 3] = 100.0
    ident = np.tile(np.array([[0.0, 0, 0, 0, 0, 0, 0, 1, 64, 64]]), (b, 1))
    xy = R.project_points(p3d.reshape(-1, kp, 3), cam, poses.reshape(-1, 3, 4)).reshape(b, k - 1, 1, kp, 2)
    batch = dict(img=torch.from_numpy(img), target_seg=torch.from_numpy(seg), keypoints3d=torch.from_numpy(p3d), target_vert=torch.from_numpy(xy[..., ::-1].copy()),
                 cam_mat=torch.from_numpy(cam), offsets=torch.from_numpy(ident), poses_gt=torch.from_numpy(poses))
    opt = SimpleNamespace(train_vectors_with_ground_truth=True, estimate_coords=True, max_keypoint_pixel_error=12.5, confidence_regularization=True,
                          use_bpnp_reprojection_loss=bpnp)  # True: config_13.ini -- host PnP + implicit gradient inside the keypoint loss
    lf = LossWeightHandler(1.0, 0.5, 0.015, 0.007, filter_vertex_with_segmentation=True)
    optim = Adam(learning_rate=PiecewiseConstantDecay([5], [1e-3, 5e-4]))
    hist = [train_step(net, batch, lf, optim, opt) for _ in range(10)]
    hist = np.array(hist)
    assert np.all(np.isfinite(hist)) and optim.iterations == 10 and optim.lr == 5e-4
    assert hist[-1, 0] < hist[0, 0]
    after = net([img, seg], training=False).cpu().numpy()
    assert np.abs(after - before).max() > 1e-3, "inference must see the trained weights"
    p = net.get_parameters()
    assert np.abs(p["conv0.kernel"] - net._store.view("conv0.kernel").cpu().numpy()).max() == 0
    # layer surgery after training has started (round-1 ADVICE): get_weights must return the TRAINED values and set_weights on one
    # layer must not revert the others to their pre-training state
    trained = net.get_parameters()
    lay = net.get_layer("pv_block_6_clade")
    wts = lay.get_weights()
    assert [a.shape for a in wts[:2]] == [(k, 256), (k, 256)] and np.array_equal(wts[0], trained["pv_block_6_clade.beta"])   # Keras order: beta, gamma
    assert np.array_equal(net.get_layer("stage1_unit1_conv1").get_weights()[0], trained["stage1_unit1_conv1.kernel"])
    train_step(net, batch, lf, optim, opt)
    trained = net.get_parameters()
    lay.set_weights(lay.get_weights())
    again = net.get_parameters()
    assert all(np.array_equal(trained[n], again[n]) for n in trained)
    assert np.abs(again["conv0.kernel"] - net._store.view("conv0.kernel").cpu().numpy()).max() == 0
    # frozen layers (layer.trainable = False) keep their weights
    net.get_layer("conv0").trainable = False
    w0 = net.get_parameters()["conv0.kernel"].copy()
    w1 = net.get_parameters()["stage1_unit1_conv1.kernel"].copy()
    train_step(net, batch, lf, optim, opt)
    assert np.array_equal(net.get_parameters()["conv0.kernel"], w0) and not np.array_equal(net.get_parameters()["stage1_unit1_conv1.kernel"], w1)
    # evaluation-mode step (train=False): no update
    it = optim.iterations
    ev = train_step(net, batch, lf, optim, opt, train=False)
    assert optim.iterations == it and np.isfinite(ev).all()


def test_weight_surgery_helpers(device):
    """copy_weights_from_backup_network / copy_weights_add_confidence_maps (train_casapose.py:352-448) on the Keras-like layer API."""
    from casapose_amd.pose_models.tfkeras import Classifiers
    from casapose_amd.training import copy_weights_add_confidence_maps, copy_weights_from_backup_network

    mk = lambda k, v, seed: Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=(32, 32, 3), weights=None, device=device, seed=seed)  # noqa: E731
    old, new = mk(4, 27, 1), mk(6, 27, 2)
    table = np.array([[0, 0], [1, 3], [3, 5]])     # background, old object 1 -> new class 3, old object 3 -> new class 5
    before = new.get_parameters()
    copy_weights_from_backup_network(new, old, table, print_fn=lambda *_: None)
    po, pn = old.get_parameters(), new.get_parameters()
    assert np.array_equal(pn["pv_final_conv_segmentation.kernel"][0, 0, :, [0, 3, 5]], po["pv_final_conv_segmentation.kernel"][0, 0, :, [0, 1, 3]])
    assert np.array_equal(pn["pv_final_conv_segmentation.kernel"][..., [1, 2, 4]], before["pv_final_conv_segmentation.kernel"][..., [1, 2, 4]])
    for i in range(6, 11):
        for f in ("gamma", "beta"):
            key = "pv_block_%d_clade.%s" % (i, f)
            assert np.array_equal(pn[key][[0, 3, 5]], po[key][[0, 1, 3]]) and np.array_equal(pn[key][[1, 2, 4]], before[key][[1, 2, 4]])
    assert np.array_equal(pn["conv0.kernel"], before["conv0.kernel"])
    noconf, conf = mk(4, 18, 3), mk(4, 27, 4)
    b2 = conf.get_parameters()["pv_final_conv_vertex.kernel"].copy()
    copy_weights_add_confidence_maps(conf, noconf, 18, print_fn=lambda *_: None)
    k2 = conf.get_parameters()["pv_final_conv_vertex.kernel"]
    assert np.array_equal(k2[..., :18], noconf.get_parameters()["pv_final_conv_vertex.kernel"]) and np.array_equal(k2[..., 18:], b2[..., 18:])


# --------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: the 13-object network (config_13.ini) with use_bpnp_reprojection_loss = 1, at the training size
# --------------------------------------------------------------------------------------------------
def test_config13_bpnp_step_at_448_matches_autograd(device):
    """K = 14, 448x448, one image of the ray-cast 13-object scene: forward with batch statistics, mask / vertex / proxy losses and the
    keypoint loss in its BPnP form (host PnP + implicit-function gradient, loss_functions.py:264-323, config_13.ini) TOGETHER, then
    the whole backward.  Reference: oracle/torch_train_ref.py in fp64 -- autograd through the network and the differentiable LS
    voter, with d loss / d keypoints of the BPnP term supplied by the same host routine evaluated on the REFERENCE's voted
    keypoints (that routine's implicit gradient is itself checked against finite differences below, at this configuration).
    Gates: outputs <= 1e-3 of range, loss values <= 1e-3 relative, every trainable variable's gradient <= 1e-3 relative L2 against the oracle on
    the device's activation branches (which may differ from the oracle's own only at |pre-activation| < 1e-4)."""
    from casapose_amd import training as T
    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset
    from casapose_amd.train_engine import ParamStore, TrainPlan, crop_to_image_affine, project_keypoints

    b, h, w, k, kp, v = 1, 448, 448, 14, 9, 27
    ds = SyntheticSceneDataset(k - 1, (h, w), length=1, seed=6)
    batch = ds.batch(0, b)
    img = batch["img"].numpy().astype(np.float32)
    lab = batch["filtered_seg"][..., 0].numpy().astype(np.uint8)
    kpts = batch["target_vert"][:, :, 0].numpy().astype(np.float32)                                  # [b,oc,kp,2] (y,x) crop pixels
    params = O.init_params(k, v, seed=13, dtype=np.float32)
    store = ParamStore(params, device)
    plan = TrainPlan(store, k, v, b, h, w)
    plan.refresh_weights(torch.cuda.current_stream(device).cuda_stream)
    labd, kd = torch.from_numpy(lab).to(device), torch.from_numpy(kpts).to(device)
    out = plan.forward(torch.from_numpy(img).to(device), cond_labels=labd).cpu().numpy()
    # ---- reference forward ----
    p64 = R.to_torch(params)
    labt = torch.from_numpy(lab.astype(np.int64))
    pre = {}
    ref = R.forward_train(p64, torch.from_numpy(img.astype(np.float64)), labt, preact_out=pre)
    assert rel(out[..., :k], ref.detach().numpy()[..., :k]) < 1e-3 and rel(out[..., k:], ref.detach().numpy()[..., k:]) < 1e-3
    # ---- losses: mask + vertex + proxy on both sides ----
    wts, kp_w, cap = (1.0, 0.5, 0.015), 0.007, 12.5
    sums = plan.loss_and_grad(labd, labd, kd, *wts, filter_with_segmentation=True).cpu().numpy()
    ml, vl, pl = R.losses(ref, labt, torch.from_numpy(kpts.astype(np.float64)), k, kp, True)
    for got, want in zip(sums[:3], (ml, vl, pl)):
        assert abs(got - want.item()) < 1e-3 * abs(want.item())
    # ---- keypoint loss, BPnP form ----
    cam = batch["cam_mat"].numpy().astype(np.float64)
    cam = cam[0] if cam.ndim == 3 else cam
    p3d = batch["keypoints3d"].numpy().astype(np.float64).reshape(b, k - 1, kp, 3)
    gt_xy_np = project_keypoints(p3d, cam, batch["poses_gt"].numpy().astype(np.float64).reshape(b, k - 1, 3, 4))
    aff_np = crop_to_image_affine(batch["offsets"].numpy().astype(np.float64))
    gt_xy, aff = torch.from_numpy(gt_xy_np).to(device).contiguous(), torch.from_numpy(aff_np).to(device).contiguous()
    seen = {}

    def host_loss(c, av):
        lv, g, _ = T.bpnp_reprojection_loss_host(c, gt_xy, aff, av, p3d, cam, cap, kp_w, rng=np.random.default_rng(5))   # same RANSAC draws on both sides
        seen["coords"], seen["avail"], seen["loss"], seen["g"] = T._host(c).copy(), T._host(av).copy(), lv, g
        return lv, g

    val = plan.kp_loss_and_grad(labd, gt_xy, aff, kp_w, max_pixel_error=cap, min_num=50, confidence_regularization=True, vote_with_gt=True, host_loss=host_loss)
    coords_ref = R.ls_voting(labt, ref[..., k:k + 2 * kp], ref[..., k + 2 * kp:], k - 1)
    avail = seen["avail"]
    assert avail.sum() >= 6, "the scene must show enough objects for the test to mean something"
    big = avail[0] > 0
    assert np.abs(seen["coords"][0][big] - coords_ref.detach().numpy()[0][big]).max() < 0.05 + 1e-3 * np.abs(coords_ref.detach().numpy()[0][big]).max()
    # the host PnP (RANSAC inlier selection) is not continuous in its input: evaluated at the reference's keypoints it can pick another inlier
    # set than at the GPU's (they agree to < 0.05 px, asserted above) and return a gradient that differs by several per cent -- seen with
    # CASAPOSE_CONV_MODE=f32.  The host routine is therefore evaluated ONCE, at the keypoints the GPU voted; what this test checks is the
    # chain behind it (voter and network backward); the routine's own gradient is checked against finite differences below.
    lv_ref, g_ref, _ = T.bpnp_reprojection_loss_host(seen["coords"], gt_xy_np, aff_np, avail, p3d, cam, cap, kp_w, rng=np.random.default_rng(5))
    # confidence regulariser |mean_fg softplus(conf) - 0.7| (loss_functions.py:325-342) in torch
    fg = (labt > 0).double()
    cl = (F.softplus(ref[..., k + 2 * kp:]) * fg[..., None]).sum((1, 2)) / fg.sum((1, 2))[:, None]
    reg = (cl - 0.7).abs().mean()
    assert abs(val.item() - (lv_ref + reg.item())) < 2e-3 * abs(lv_ref + reg.item())
    plan.backward()
    torch.cuda.synchronize()
    # gradients against the oracle evaluated on the DEVICE's activation branches (see test_train_forward_backward_matches_autograd)
    pattern = plan.activation_pattern()
    flips, total_el, margin = R.kink_report(pattern, pre)
    assert margin < 1e-4 and flips < 1e-4 * total_el, (flips, total_el, margin)
    if flips:
        p64 = R.to_torch(params)
        ref = R.forward_train(p64, torch.from_numpy(img.astype(np.float64)), labt, act_pattern=pattern)
        ml, vl, pl = R.losses(ref, labt, torch.from_numpy(kpts.astype(np.float64)), k, kp, True)
        coords_ref = R.ls_voting(labt, ref[..., k:k + 2 * kp], ref[..., k + 2 * kp:], k - 1)
        cl = (F.softplus(ref[..., k + 2 * kp:]) * fg[..., None]).sum((1, 2)) / fg.sum((1, 2))[:, None]
        reg = (cl - 0.7).abs().mean()
    total = wts[0] * ml + wts[1] * vl + wts[2] * pl + kp_w * reg + (coords_ref * torch.from_numpy(g_ref.astype(np.float64))).sum()
    total.backward()
    worst = {name: rel_l2(store.grad_view(name).cpu().numpy(), p64[name].grad.numpy()) for name in store.offsets}
    top = sorted(worst.items(), key=lambda t: -t[1])
    print("config 13 @448: %d kink flips (margin %.1e); worst gradient %s %.2e, median %.2e" % (flips, margin, top[0][0], top[0][1], np.median(list(worst.values()))))
    # 1e-3 in the plan's default arithmetic (exact splits: measured median 6e-5).  With CASAPOSE_CONV_MODE=f32 CASAPOSE_WINO_GEMM=f32 every product is
    # rounded by the fp32 MFMA and this configuration lands at median 3e-3 / worst 4.4e-3 WITHOUT any kernel being wrong: the proxy-voting
    # term divides by |v|^2 of a random-initialised direction field, so a few pixels with |v| ~ 1e-3 dominate the gradient and turn the
    # forward's 1e-5 into per cents (tools/debug/grad_per_variable.py at 224x224: 8e-3 worst with fp32 MFMA and direct kernels only, 3.6e-3
    # with the exact splits) -- that mode is gated at 1e-2 here and at 1e-3 on the better-conditioned shapes of the tests above.
    # CASAPOSE_TRAIN_FWD=f16x2 (opt-in: the forward in the fp16 two-way split, fp32-LEVEL rather than exact) lands where the fp32 MFMA does: 4.4e-3 worst.
    # Round 6: that fp32-level forward is the plan's DEFAULT (its operand range is watched on the device, train_engine.TrainPlan._poll_f16x2);
    # CASAPOSE_TRAIN_FWD=split restores the exact forward and with it the 1e-3 gate.
    exact = os.environ.get("CASAPOSE_CONV_MODE", "split") == "split" and os.environ.get("CASAPOSE_TRAIN_FWD", "f16x2") != "f16x2"
    gate = 1e-3 if exact else 1e-2
    bad = {n: e for n, e in worst.items() if e > gate}
    assert not bad, "gradient mismatch (relative L2): %s" % sorted(bad.items(), key=lambda t: -t[1])[:10]
    # ---- the host BPnP gradient itself, by central differences on two visible objects of THIS scene ----
    c0 = coords_ref.detach().numpy().copy()
    one = np.zeros_like(avail)
    for o in np.nonzero(big)[0][:2]:
        one[:] = 0
        one[0, o] = 1
        f = lambda c: T.bpnp_reprojection_loss_host(c, gt_xy_np, aff_np, one, p3d, cam, cap, 1.0, rng=np.random.default_rng(5))  # noqa: E731  same RANSAC draws every call
        _, g, _ = f(c0)
        for j, a in ((0, 0), (4, 1), (8, 0)):
            cp, cm = c0.copy(), c0.copy()
            cp[0, o, j, a] += 1e-3
            cm[0, o, j, a] -= 1e-3
            fd = (f(cp)[0] - f(cm)[0]) / 2e-3
            assert abs(fd - g[0, o, j, a]) < 2e-3 * max(1.0, np.abs(g[0, o]).max()) + 2e-2 * abs(fd), (o, j, a, fd, g[0, o, j, a])


def test_f16x2_forward_monitor_moves_an_out_of_band_layer_to_the_exact_split(device, monkeypatch):
    """Round 6: the training plan's forward runs in the fp16 two-way split by default, and every such launch reports max |x| of what it converts into a
    device-side monitor slot (cp_f16x2_monitor_set).  A normalisation whose gamma is 1e5 puts the input of ONE convolution far beyond the fp16 range:
    the first forward clamps there (its output is wrong by per cents), the monitor sees it without a synchronisation in the step, that op's forward
    moves to the exact three-way split -- the others stay -- and the plan's output then agrees with an all-exact-forward plan to fp32 level."""
    import warnings

    from casapose_amd import train_engine as TE

    monkeypatch.setattr(TE, "F16X2_TRAIN_CHECK_EVERY", 1)
    monkeypatch.delenv("CASAPOSE_TRAIN_FWD", raising=False)
    monkeypatch.delenv("CASAPOSE_CONV_MODE", raising=False)
    k, v, b, h, w = 4, 27, 2, 32, 48
    params = O.init_params(k, v, seed=21, dtype=np.float32)
    params["stage1_unit1_bn2.gamma"] = (params["stage1_unit1_bn2.gamma"] * 1e5).astype(np.float32)   # -> the input of stage1_unit1_conv2 peaks near 1e5
    rng = np.random.default_rng(2)
    img = torch.from_numpy(rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)).to(device)
    lab = np.zeros((b, h, w), np.uint8)
    lab[:, 4:20, 6:30] = 1
    lab[:, 14:30, 20:44] = 2
    labd = torch.from_numpy(lab).to(device)

    def plan_for(fwd):
        if fwd:
            monkeypatch.setenv("CASAPOSE_TRAIN_FWD", fwd)
        else:
            monkeypatch.delenv("CASAPOSE_TRAIN_FWD", raising=False)
        plan = TE.TrainPlan(TE.ParamStore(params, device), k, v, b, h, w)
        plan.refresh_weights(torch.cuda.current_stream(device).cuda_stream)
        return plan

    exact = plan_for("split")
    ref = exact.forward(img, cond_labels=labd).clone()
    plan = plan_for(None)
    assert any(isinstance(op, TE.ConvOp) and op.layer.fwd_f16x2 for op in plan.ops) and not any(isinstance(op, TE.ConvOp) and op.layer.fwd_f16x2 for op in exact.ops)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        first = plan.forward(img, cond_labels=labd).clone()    # arms the slots (nothing to judge yet)
        plan.forward(img, cond_labels=labd)                    # reports; the slots travel to the host behind this step
        torch.cuda.synchronize()
        last = plan.forward(img, cond_labels=labd).clone()     # judged at its start: the op is demoted before it runs
        torch.cuda.synchronize()
    assert plan.f16x2_checks >= 1 and any(n.startswith("stage1_unit1_conv2") for n in plan.f16x2_demoted), plan.f16x2_demoted
    assert len(plan.f16x2_demoted) <= 3, plan.f16x2_demoted   # the layer itself (and at most what reads the same tensor), not the plan
    assert any("exact bf16 split" in str(c.message) for c in caught)
    still = [op.layer.name for op in plan.ops if isinstance(op, TE.ConvOp) and op.layer.fwd_f16x2]
    assert len(still) >= 20 and "stage1_unit1_conv2" not in still, still
    scale = float(ref.abs().max())
    e_first, e_last = float((first - ref).abs().max()) / scale, float((last - ref).abs().max()) / scale
    print("error against the exact forward: clamping f16x2 forward %.2e, after the demotion %.2e" % (e_first, e_last))
    assert e_last <= 1e-4 and e_first > 10 * e_last, (e_first, e_last)


@pytest.mark.gpu
def test_backward_in_the_fp16_two_way_split_matches_the_exact_split(device, monkeypatch):
    """Round 6: the backward GEMMs of the 3x3 layers in the fp16 two-way split (train_engine.train_bwd_f16x2).  The first backward of a plan runs on
    the exact split and measures; from the second on the Winograd data / weight gradients carry their own power of two, the loss carries one for the
    direct layers' data gradients, and the flat gradient is taken back by it before anything reads it.  With identical parameters and inputs the
    gradient of the SECOND step-free backward must agree with an all-exact plan's to fp32 level, GEMMs must actually have moved, and a loss 1000
    times smaller must be followed by the exponents (no synchronisation) with the same agreement."""
    from casapose_amd import engine as E
    from casapose_amd import train_engine as TE

    if not E.TRAIN_WINO_GEMM_SPLIT:
        pytest.skip("CASAPOSE_WINO_GEMM=f32: the Winograd GEMMs of this process run on the fp32 MFMA (read at import), there is no fp16-pair backward to test")
    monkeypatch.setattr(TE, "F16X2_TRAIN_CHECK_EVERY", 1)
    for v_ in ("CASAPOSE_TRAIN_FWD", "CASAPOSE_CONV_MODE", "CASAPOSE_TRAIN_BWD"):
        monkeypatch.delenv(v_, raising=False)
    k, v, b, h, w = 4, 27, 2, 96, 128
    params = O.init_params(k, v, seed=5, dtype=np.float32)
    rng = np.random.default_rng(3)
    img = torch.from_numpy(rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32)).to(device)
    lab = np.zeros((b, h, w), np.uint8)
    lab[:, 10:60, 12:70] = 1
    lab[:, 40:90, 60:120] = 2
    lab[:, 5:30, 90:125] = 3
    labd = torch.from_numpy(lab).to(device)
    kpts = torch.from_numpy(rng.uniform(0, min(h, w), (b, k - 1, 9, 2)).astype(np.float32)).to(device)

    def plan_for(bwd):
        monkeypatch.setenv("CASAPOSE_TRAIN_BWD", bwd)
        plan = TE.TrainPlan(TE.ParamStore(params, device), k, v, b, h, w)
        plan.refresh_weights(torch.cuda.current_stream(device).cuda_stream)
        return plan

    def grad_of(plan, wts):
        plan.forward(img, cond_labels=labd)
        plan.loss_and_grad(labd, labd, kpts, *wts)
        plan.backward()
        torch.cuda.synchronize()
        return plan.store.grad.double().cpu().numpy().copy()

    def rel(a, b_):
        return float(np.abs(a - b_).max() / np.abs(b_).max())

    wts = (1.0, 0.5, 0.015)
    exact = plan_for("split")
    ref = grad_of(exact, wts)
    plan = plan_for("f16x2")
    g0 = grad_of(plan, wts)          # exact split, measuring; calibrated synchronously at its end
    assert rel(g0, ref) < 2e-5
    slots = plan._bwd_slots()
    kinds = {"direct": [f for _, f, e in slots if e == "direct"], "wino": [f for _, f, e in slots if e != "direct"]}
    assert kinds["direct"] and kinds["wino"]
    assert sum(1 for f in kinds["direct"] if f["on"]) >= len(kinds["direct"]) // 2, [f["on"] for f in kinds["direct"]]
    assert all(f["e"] is not None for f in kinds["wino"])
    assert plan.loss_exp > 0   # gradients of a mean loss are far below fp16's band
    g1 = grad_of(plan, wts)          # fp16 pairs
    assert rel(g1, ref) < 2e-5, rel(g1, ref)
    # per-variable view of the same comparison: every tensor of the flat gradient agrees, not only the largest
    st = plan.store
    worst = max(rel(g1[off:off + int(np.prod(shape))], ref[off:off + int(np.prod(shape))]) for _, (off, shape) in st.offsets.items()
                if np.abs(ref[off:off + int(np.prod(shape))]).max() > 0)
    assert worst < 5e-4, worst
    # a loss 1000 times smaller: the slots report it, the exponents follow at the start of a later step
    small = tuple(x * 1e-3 for x in wts)
    e_before = plan.loss_exp
    for _ in range(3):
        gs = grad_of(plan, small)
    assert plan.loss_exp >= e_before + 8, (e_before, plan.loss_exp)
    refs = grad_of(exact, small)
    assert rel(gs, refs) < 2e-5, rel(gs, refs)
