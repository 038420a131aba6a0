"""cp_conv2d_fwd_bf16_deep (csrc/conv_bf16d.hip): the direct bf16-operand convolution for the deep 3x3 layers (BASELINE configs[2]) against an fp64
convolution of the SAME bf16-rounded operands -- products of bf16 values are exact in fp32, so what is left is the fp32 accumulation order
(gate 2e-5 of the output range) -- and against the fp64 convolution of the unrounded operands at the bf16 mode's 3e-2 gate."""
import ctypes as C

import numpy as np
import pytest
import torch

import torch_train_ref as R

pytestmark = pytest.mark.gpu


def bf16_round(a):
    """round to nearest even to 8 significand bits, as the kernel's round4()"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)


CASES = [
    # name, dilation, sources [(channels)], cout, (b, h, w), residual, act
    ("d1_two_sources", 1, [64, 32], 128, (2, 20, 40), False, 0),
    ("d2_residual_relu", 2, [48], 256, (2, 18, 33), True, 1),
    ("d4_wide_tile", 4, [32], 128, (1, 14, 70), True, 2),      # 8 x 64 tiles (70 columns: 128 of 64 against 96 of 32 ... both shapes get run over the cases)
    ("d4_tall_tile", 4, [32], 128, (1, 37, 30), False, 0),     # 16 x 32 tiles, ragged rows
    ("d1_stage4_like", 1, [128], 384, (1, 16, 32), False, 1),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_bf16_deep_conv_matches_fp64_on_rounded_operands(device, hip_lib, case):
    lib = hip_lib
    from casapose_amd import _lib
    from casapose_amd._lib import ConvDesc, check

    name, dil, chans, cout, (b, h, w), use_res, act = case
    rng = np.random.default_rng(sum(name.encode()))
    ns = len(chans)
    cin = sum(chans)
    xs = [rng.standard_normal((b, h, w, c)).astype(np.float32) for c in chans]
    wk = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    res = rng.standard_normal((b, h, w, cout)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    st = torch.cuda.current_stream(device).cuda_stream
    # weights: conv_hsplit's fragment stream (64-channel passes), one bf16 plane
    ch = (C.c_int * 2)(*(chans + [0] * (2 - ns)))
    nfl = lib.cp_conv_split_weight_floats(cout, ns, ch)
    packed = np.zeros(nfl, np.float32)
    check(lib.cp_conv_pack_weights_split_host(wk.ctypes.data, 0, cout, ns, ch, ch, packed.ctypes.data))
    pk = torch.from_numpy(packed).to(device)
    planes = torch.empty(nfl // 512 * 1024, dtype=torch.uint8, device=device)
    check(lib.cp_conv_split_weights_f32(pk.data_ptr(), nfl, 1, planes.data_ptr(), st))
    xd = [torch.from_numpy(x).to(device) for x in xs]
    resd, scd, shd = torch.from_numpy(res).to(device), torch.from_numpy(scale).to(device), torch.from_numpy(shift).to(device)
    raw = torch.full((b, h, w, cout), 7.0, device=device)
    actt = torch.full((b, h, w, cout), 7.0, device=device)
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w = b, h, w, h, w
    d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = cout, 3, 3, 1, dil, dil
    d.num_sources = ns
    for i, c in enumerate(chans):
        d.src[i].data, d.src[i].channels, d.src[i].ld, d.src[i].mode = xd[i].data_ptr(), c, c, _lib.SRC_DIRECT
    d.residual, d.residual_ld = (resd.data_ptr() if use_res else None), cout
    d.out_raw, d.out_raw_ld = raw.data_ptr(), cout
    if act:
        d.out_act, d.out_act_ld, d.scale, d.shift, d.act = actt.data_ptr(), cout, scd.data_ptr(), shd.data_ptr(), act
    assert lib.cp_conv_bf16_deep_applicable(C.byref(d)) == 1
    check(lib.cp_conv2d_fwd_bf16_deep(C.byref(d), planes.data_ptr(), st))
    torch.cuda.synchronize()
    x64 = torch.from_numpy(np.concatenate([bf16_round(x) for x in xs], -1).astype(np.float64))
    ref = R.conv_nhwc(x64, torch.from_numpy(bf16_round(wk).astype(np.float64)), stride=1, dilation=dil, pad=dil).numpy()
    if use_res:
        ref = ref + res
    got = raw.cpu().numpy()
    assert rel(got, ref) < 2e-5, "raw output vs fp64 on the rounded operands"
    exact = R.conv_nhwc(torch.from_numpy(np.concatenate(xs, -1).astype(np.float64)), torch.from_numpy(wk.astype(np.float64)), stride=1, dilation=dil, pad=dil).numpy()
    assert rel(got, exact + (res if use_res else 0)) < 3e-2, "the bf16 mode's gate against the unrounded convolution"
    if act:
        t = ref * scale + shift
        want = np.maximum(t, 0) if act == 1 else np.maximum(t, 0) - np.maximum(-0.1 * t, 0)
        assert rel(actt.cpu().numpy(), want) < 2e-5
    else:
        assert float(actt.min()) == 7.0   # untouched


def test_bf16_deep_refuses_what_it_does_not_cover(device, hip_lib):
    from casapose_amd import _lib
    from casapose_amd._lib import ConvDesc

    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w = 1, 8, 8, 8, 8
    d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = 64, 3, 3, 1, 1, 1    # cout not a multiple of 128
    d.num_sources = 1
    x = torch.zeros(1, 8, 8, 32, device=device)
    o = torch.zeros(1, 8, 8, 64, device=device)
    d.src[0].data, d.src[0].channels, d.src[0].ld, d.src[0].mode = x.data_ptr(), 32, 32, _lib.SRC_DIRECT
    d.out_raw, d.out_raw_ld = o.data_ptr(), 64
    assert hip_lib.cp_conv_bf16_deep_applicable(C.byref(d)) == 0
    assert hip_lib.cp_conv2d_fwd_bf16_deep(C.byref(d), x.data_ptr(), None) != 0
