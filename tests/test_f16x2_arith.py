"""CPU: the arithmetic claims of the fp16 two-way split (casapose_amd/csrc/split_f16.h) restated in NumPy -- no GPU, no library
compute.  hi = rn_f16(x), lo = rn_f16(x - hi):
  * the pair reproduces x to within ONE fp32 ulp (2^-23 |x|; rms 0.75 * 2^-24, three operands in four exactly) while lo is a normal fp16 number
    (|x| >= 2^-2), to an absolute 2^-24 below (fp16 subnormals);
  * hi*hi, hi*lo, lo*hi are exact in fp32 (11 x 11-bit significands), and dropping lo*lo costs <= 2^-22 of the product;
  * a dot product formed that way in fp32 is as close to the fp64 result as an fp32 dot product of the unsplit operands;
  * weights need the power-of-two scale (cp_f16x2_weight_scale) for that to hold when they are small.
The GPU kernels are held to the same statements in tests/test_gpu_f16x2.py."""
import numpy as np
import pytest


def split(x):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def test_the_pair_reproduces_an_fp32_operand_to_half_an_ulp():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * np.exp2(rng.uniform(-2, 15.9, 200000))).astype(np.float32)
    x = x[(np.abs(x) >= 0.25) & (np.abs(x) <= 65504.0)]
    hi, lo = split(x)
    assert np.all(np.isfinite(hi.astype(np.float32)))
    r = x.astype(np.float64) - hi.astype(np.float64)
    assert np.all(r.astype(np.float32).astype(np.float64) == r)              # x - hi is exact in fp32
    err = np.abs(x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    rel = err / np.abs(x.astype(np.float64))
    # x - hi has up to 13 significant bits and lo keeps 11 of them: at most one unit of the LAST place of x is lost
    assert rel.max() <= 2.0 ** -23 and np.sqrt(np.mean(rel ** 2)) <= 0.8 * 2.0 ** -24 and np.mean(err == 0) > 0.7, (rel.max() * 2 ** 24, np.mean(err == 0))
    small = (rng.standard_normal(100000) * np.exp2(rng.uniform(-20, -2, 100000))).astype(np.float32)
    hi, lo = split(small)
    err = np.abs(small.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    assert np.all(err <= 2.0 ** -24)                                          # subnormal low parts: absolute (half the fp16 subnormal spacing 2^-24 ... plus rn), not relative


def test_the_three_products_are_exact_in_fp32_and_the_dropped_one_is_below_fp32_rounding():
    rng = np.random.default_rng(2)
    a = (rng.standard_normal(100000) * 8).astype(np.float32)
    b = (rng.standard_normal(100000) * 2048).astype(np.float32)
    (ah, al), (bh, bl) = split(a), split(b)
    for p, q in ((ah, bh), (ah, bl), (al, bh)):
        exact = p.astype(np.float64) * q.astype(np.float64)
        assert np.all((p.astype(np.float32) * q.astype(np.float32)).astype(np.float64) == exact)
    full = a.astype(np.float64) * b.astype(np.float64)
    three = ah.astype(np.float64) * bh.astype(np.float64) + ah.astype(np.float64) * bl.astype(np.float64) + al.astype(np.float64) * bh.astype(np.float64)
    big = (np.abs(a) >= 0.25) & (np.abs(b) >= 0.25)
    rel = (np.abs(three - full) / np.abs(full))[big]
    assert rel.max() <= 2.0 ** -21 and np.sqrt(np.mean(rel ** 2)) <= 2.0 ** -23, (rel.max(), np.sqrt(np.mean(rel ** 2)))


@pytest.mark.parametrize("k,a_scale,w_scale", [(576, 1.0, 0.05), (4608, 3.0, 0.02), (512, 30.0, 1.0)])
def test_a_split_dot_product_is_as_good_as_an_fp32_one(k, a_scale, w_scale):
    from casapose_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(k)
    a = np.maximum(rng.standard_normal((64, k)) * a_scale, 0).astype(np.float32)
    w = (rng.standard_normal((32, k)) * w_scale).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    den = np.abs(ref).max()
    f32 = (a @ w.T).astype(np.float64)                                        # an fp32 dot product (BLAS order)
    s = float(lib.cp_f16x2_weight_scale(float(np.abs(w).max())))
    assert 2048.0 <= float(np.abs(w).max()) * s < 4096.0 and np.log2(s) == np.round(np.log2(s))
    (ah, al), (wh, wl) = split(a), split(w * np.float32(s))
    f = lambda t: t.astype(np.float32)
    got = ((f(ah) @ f(wh).T) + (f(ah) @ f(wl).T) + (f(al) @ f(wh).T)).astype(np.float64) / s   # fp32 accumulation of exact products
    e_split, e_f32 = np.abs(got - ref).max() / den, np.abs(f32 - ref).max() / den
    assert e_split <= 2.0 * e_f32 + 1e-7, (e_split, e_f32)
    # without the weight scale the low parts of small weights are subnormal and the representation error of the dot product grows (isolated with
    # exact accumulation: the measured 8.9e-8 -> 2.0e-8 rms of DESIGN.md 4.1f)
    d = lambda t: t.astype(np.float64)
    (wh0, wl0) = split(w)
    rms = lambda e: float(np.sqrt(np.mean(e ** 2))) / den
    rep_scaled = rms((d(ah) @ d(wh).T + d(ah) @ d(wl).T + d(al) @ d(wh).T) / s - ref)
    rep_plain = rms(d(ah) @ d(wh0).T + d(ah) @ d(wl0).T + d(al) @ d(wh0).T - ref)
    assert rep_scaled <= 4e-8, rep_scaled
    if w_scale < 0.1:
        assert rep_plain > 2.0 * rep_scaled, (rep_plain, rep_scaled)


def test_training_weight_scale_is_cached_between_rescale_points(monkeypatch):
    """train_engine.f16x2_scale: max |w| needs the host, so the power-of-two scale of the opt-in f16x2 forward is re-read every
    F16X2_RESCALE_EVERY refreshes and kept (16x headroom) in between."""
    import torch

    from casapose_amd import train_engine as te

    monkeypatch.setattr(te, "F16X2_RESCALE_EVERY", 4)
    cache = {}
    w = torch.full((8,), 0.03)
    s0 = te.f16x2_scale(cache, w)
    assert 2048.0 <= 0.03 * s0 < 4096.0
    w.mul_(3.0)                                   # weights move between rescale points: the scale stays
    assert [te.f16x2_scale(cache, w) for _ in range(3)] == [s0, s0, s0]
    s4 = te.f16x2_scale(cache, w)                 # fifth call = refresh 4: re-read
    assert 2048.0 <= 0.09 * s4 < 4096.0 and s4 == s0 / 4.0
