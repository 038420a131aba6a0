"""CPU checks of the C-ABI library: it builds for gfx950 without a GPU, loads, exports every
symbol that include/casapose_hip.h declares, and its host-side helpers (K layout, weight
packing, argument validation) behave as documented.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from casapose_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "casapose_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from casapose_amd import _lib

    names = declared_functions()
    assert len(names) >= 18
    bound = {n for n, _, _ in _lib.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n
        assert n in bound, "%s is exported but casapose_amd/_lib.py does not bind it" % n
    assert bound <= set(names), "bound symbols missing from the header: %s" % (bound - set(names))


def test_version_and_device_probe(lib):
    assert lib.cp_version() >= 100
    assert lib.cp_device_count() >= 0  # 0 on the CPU-only build container; never raises


def test_ktot_layout(lib):
    ch = (C.c_int * 2)
    assert lib.cp_conv_ktot(3, 3, 1, ch(64, 0)) == 9 * 64
    assert lib.cp_conv_ktot(7, 7, 1, ch(4, 0)) == 7 * 32          # 49 taps x 4 ch, 8 taps per 32-wide chunk
    assert lib.cp_conv_ktot(3, 3, 2, ch(32, 4)) == 9 * 32 + 2 * 32
    assert lib.cp_conv_ktot(1, 1, 1, ch(512, 0)) == 512


def test_weight_packing_orders_k_by_source_tap_channel(lib):
    kh = kw = 3
    c0, c1r, cout = 32, 3, 5
    w = np.arange(kh * kw * (c0 + c1r) * cout, dtype=np.float32).reshape(kh, kw, c0 + c1r, cout)
    chans, real = (C.c_int * 2)(c0, 4), (C.c_int * 2)(c0, c1r)
    ktot = lib.cp_conv_ktot(kh, kw, 2, chans)
    dst = np.full((cout, ktot), -1.0, np.float32)
    assert lib.cp_conv_pack_weights_host(w.ctypes.data, 0, kh, kw, cout, 2, chans, real, dst.ctypes.data) == 0
    for co in (0, 4):
        for t in range(9):
            ky, kx = divmod(t, 3)
            assert np.array_equal(dst[co, t * 32 : (t + 1) * 32], w[ky, kx, :32, co])      # source 0: tap-major
            base = 9 * 32 + t * 4
            assert np.array_equal(dst[co, base : base + 3], w[ky, kx, 32:35, co])           # source 1: 4 per tap
            assert dst[co, base + 3] == 0.0                                                   # padded channel
    assert (dst[:, 9 * 32 + 36 :] == 0).all()                                                # K padding
    # IHWO layout (PartialConvolution.conv_w) packs to the same rows
    dst2 = np.empty_like(dst)
    w_ihwo = np.ascontiguousarray(np.transpose(w, (2, 0, 1, 3)))
    assert lib.cp_conv_pack_weights_host(w_ihwo.ctypes.data, 1, kh, kw, cout, 2, chans, real, dst2.ctypes.data) == 0
    assert np.array_equal(dst, dst2)


def test_argument_validation_reports_errors_without_a_gpu(lib):
    from casapose_amd._lib import ConvDesc

    assert lib.cp_conv2d_fwd_f32(None, None) == -1
    assert b"null descriptor" in lib.cp_last_error()
    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.cout = 1, 8, 8, 8, 8, 4
    d.kh = d.kw = 3
    d.stride = d.dilation = 1
    d.pad = 1
    d.num_sources = 1
    d.src[0].channels = 24  # neither 4 nor a multiple of 32
    d.src[0].ld = 24
    d.src[0].data = 16
    d.weights = 16
    d.out_raw = 16
    d.out_raw_ld = 4
    assert lib.cp_conv2d_fwd_f32(C.byref(d), None) == -1
    assert b"channels must be 4 or a multiple of 32" in lib.cp_last_error()
    d.out_h = 7  # inconsistent geometry
    d.src[0].channels = d.src[0].ld = 32
    assert lib.cp_conv2d_fwd_f32(C.byref(d), None) == -1
    assert b"inconsistent" in lib.cp_last_error()
    assert lib.cp_ls_vote_f32(None, 36, 0, 9, 27, None, 1, 8, 8, 8, 9, None, None, None) == -1
    assert lib.cp_argmax_labels(16, 4, 9, 10, 16, None) == -1  # ld < classes
    assert lib.cp_ls_vote_workspace_bytes(2, 8, 9) == 2 * 8 * 9 * 5 * 8
