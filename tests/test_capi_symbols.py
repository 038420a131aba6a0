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


def header_struct_fields(name):
    """[(field, kind)] of `typedef struct <name> { ... } <name>;` in the header, in declaration order; kind = "ptr", "int", "u32" or
    "<struct> * n"."""
    text = open(os.path.join(ROOT, "include", "casapose_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const )?(\w+)\s*(.*)", decl)
        base, rest = m.group(2), m.group(3)
        for item in rest.split(","):
            item = item.strip()
            arr = re.match(r"(\w+)\[(\d+)\]", item)
            if arr:
                out.append((arr.group(1), "%s * %s" % (base, arr.group(2))))
            elif item.startswith("*") or base.endswith("*"):
                out.append((item.lstrip("* "), "ptr"))
            else:
                out.append((item, {"int": "int", "uint32_t": "u32"}[base]))
    return out


def ctypes_struct_fields(cls):
    kinds = {C.c_void_p: "ptr", C.c_int: "int", C.c_uint32: "u32"}
    out = []
    for n, t in cls._fields_:
        if isinstance(t, type) and issubclass(t, C.Array):
            out.append((n, "%s * %d" % ({"ConvSource": "cp_conv_source"}[t._type_.__name__], t._length_)))
        else:
            out.append((n, kinds[t]))
    return out


def test_descriptor_structs_match_the_header_field_for_field(lib):
    """The round-2 defect: cp_conv_desc grew at its tail while a documented binding kept the shorter struct.  The ctypes declarations must
    list the header's fields in the header's order with the header's kinds, and the library must report the same sizes."""
    from casapose_amd import _lib

    assert ctypes_struct_fields(_lib.ConvSource) == header_struct_fields("cp_conv_source")
    assert ctypes_struct_fields(_lib.ConvDesc) == header_struct_fields("cp_conv_desc")
    assert _lib.ConvDesc._fields_[0][0] == "struct_size"
    assert lib.cp_conv_desc_size() == C.sizeof(_lib.ConvDesc) and lib.cp_conv_source_size() == C.sizeof(_lib.ConvSource)
    assert _lib.ConvDesc().struct_size == C.sizeof(_lib.ConvDesc)


def test_descriptor_of_another_abi_revision_is_refused(lib):
    """A caller built against a header with a shorter (or longer) cp_conv_desc is refused by size in every entry point that takes one."""
    from casapose_amd._lib import ConvDesc

    d = ConvDesc()
    d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.cout = 1, 8, 8, 8, 8, 32
    d.kh = d.kw = 3
    d.stride = d.dilation = d.pad = 1
    d.num_sources = 1
    d.src[0].channels = d.src[0].ld = 32
    d.src[0].data = d.weights = d.out_raw = 16
    d.out_raw_ld = 32
    for size in (0, C.sizeof(ConvDesc) - 16, C.sizeof(ConvDesc) + 8):
        d.struct_size = size
        assert lib.cp_conv2d_fwd_f32(C.byref(d), None) == -1 and b"struct_size" in lib.cp_last_error()
        assert lib.cp_conv2d_fwd_split(C.byref(d), 16, None, 3, None) == -1 and b"struct_size" in lib.cp_last_error()
        assert lib.cp_conv2d_wgrad_f32(C.byref(d), 16, 32, 16, 0, None) == -1 and b"struct_size" in lib.cp_last_error()
        assert lib.cp_conv2d_wgrad_split(C.byref(d), 16, 32, 16, 0, 3, None) == -1 and b"struct_size" in lib.cp_last_error()
        assert lib.cp_conv_selected_tile(C.byref(d)) == -1
        assert lib.cp_conv_split_applicable(C.byref(d)) == 0 and lib.cp_conv_wgrad_split_applicable(C.byref(d)) == 0
    d.struct_size = C.sizeof(ConvDesc)
    assert lib.cp_conv_split_applicable(C.byref(d)) == 1 and lib.cp_conv_wgrad_split_applicable(C.byref(d)) == 1


def test_integration_doc_binding_is_current():
    """INTEGRATION.md section 3 shows the binding a maintainer would write; it is generated from casapose_amd/_lib.py."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration_binding.py"), "--check"])
    assert r.returncode == 0, "INTEGRATION.md's ctypes block is stale: run python tools/gen_integration_binding.py"


def test_version_and_device_probe(lib):
    from casapose_amd import _lib

    assert lib.cp_version() == _lib.ABI_VERSION == 302
    text = open(os.path.join(ROOT, "include", "casapose_hip.h")).read()
    assert int(re.search(r"#define CP_ABI_VERSION (\d+)", text).group(1)) == _lib.ABI_VERSION
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
    assert b"descriptor is null" in lib.cp_last_error()
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


def test_f16x2_range_check_known_answers(lib):
    """cp_f16x2_range_check (round 6: the band check of the f16x2 range guard behind the C ABI): 0 inside [lo, hi] or for amax == 0, 1 with the power
    of two that brings amax into [2^10, 2^11), 2 where no power of two in [2^-24, 2^24] does or amax is not finite.  Host function: runs without a GPU."""
    lo, hi = 0.5, 65504.0 / 4.0
    r = C.c_float(7.0)
    for amax in (0.0, 0.5, 1.0, 1353.0, hi):
        assert lib.cp_f16x2_range_check(amax, lo, hi, C.byref(r)) == 0 and r.value == 1.0, amax
    for amax, want in ((0.49, 4096.0), (1e-2, 2.0 ** 17), (3e-5, 2.0 ** 26), (16377.0, 2.0 ** -3), (1e5, 2.0 ** -6), (3e3 * 100, 2.0 ** -8), (65504.0 * 4096, 2.0 ** -17)):
        st = lib.cp_f16x2_range_check(amax, lo, hi, C.byref(r))
        if want > 2.0 ** 24 or want < 2.0 ** -24:
            assert st == 2 and r.value == 1.0, (amax, st, r.value)
            continue
        assert st == 1 and r.value == want and 1024.0 <= amax * r.value < 2048.0, (amax, st, r.value)
    for bad in (float("inf"), float("nan"), -1.0, 1e-30, 3e38):
        assert lib.cp_f16x2_range_check(bad, lo, hi, C.byref(r)) == 2 and r.value == 1.0, bad
    assert lib.cp_f16x2_range_check(3.0, lo, hi, None) == 0   # the factor is optional


def test_f16x2_monitor_slot_is_per_thread_state(lib):
    import threading

    assert lib.cp_f16x2_monitor_get() is None
    assert lib.cp_f16x2_monitor_set(4096) == 0 and lib.cp_f16x2_monitor_get() == 4096
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.cp_f16x2_monitor_get()))
    t.start(); t.join()
    assert seen == [None]                      # another thread's launches are not armed
    assert lib.cp_f16x2_monitor_set(4100) == -1 and b"16-byte" in lib.cp_last_error()   # a slot is four words, 16-byte aligned
    assert lib.cp_f16x2_monitor_set(None) == 0 and lib.cp_f16x2_monitor_get() is None
