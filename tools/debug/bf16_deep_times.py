"""Timing of cp_conv2d_fwd_bf16_deep on the deep layers' shapes (bs 16, 60 x 80; bs 32, 56 x 56) beside the two-plane Winograd path it would replace."""
import ctypes as C, sys
import numpy as np, torch
sys.path[:0] = ["."]
from casapose_amd import _lib
from casapose_amd._lib import ConvDesc, check
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream

def timeit(f, n=10):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for (b, h, w) in ((16, 60, 80), (32, 56, 56)):
    for cin, cout, dil in ((512, 512, 4), (256, 512, 4), (256, 256, 2), (512, 256, 1), (384, 128, 1)):
        x = torch.randn(b, h, w, cin, device=dev)
        wk = (np.random.default_rng(0).standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
        ch = (C.c_int * 2)(cin, 0)
        nfl = lib.cp_conv_split_weight_floats(cout, 1, ch)
        packed = np.zeros(nfl, np.float32)
        check(lib.cp_conv_pack_weights_split_host(wk.ctypes.data, 0, cout, 1, ch, ch, packed.ctypes.data))
        pk = torch.from_numpy(packed).to(dev)
        planes = torch.empty(nfl // 512 * 1024, dtype=torch.uint8, device=dev)
        check(lib.cp_conv_split_weights_f32(pk.data_ptr(), nfl, 1, planes.data_ptr(), st))
        raw = torch.empty(b, h, w, cout, device=dev)
        d = ConvDesc()
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w = b, h, w, h, w
        d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = cout, 3, 3, 1, dil, dil
        d.num_sources = 1
        d.src[0].data, d.src[0].channels, d.src[0].ld, d.src[0].mode = x.data_ptr(), cin, cin, _lib.SRC_DIRECT
        d.out_raw, d.out_raw_ld = raw.data_ptr(), cout
        t = timeit(lambda: check(lib.cp_conv2d_fwd_bf16_deep(C.byref(d), planes.data_ptr(), st)))
        fl = 2.0 * b * h * w * 9 * cin * cout
        # the two-plane Winograd path on the same problem: input transform + GEMM (hi + mid planes) + output transform
        T, Tp = C.c_int(), C.c_int()
        check(lib.cp_wino_tiles(b, h, w, dil, C.byref(T), C.byref(Tp)))
        tp = Tp.value
        V = torch.zeros(36 * tp * cin, device=dev)
        M = torch.empty(36 * tp * cout, device=dev)
        U = torch.randn(36 * cout * cin, device=dev)
        Us = torch.empty(lib.cp_wino_split_weights_bytes(36, cout, cin), dtype=torch.uint8, device=dev)
        check(lib.cp_wino_split_weights_f32(U.data_ptr(), 36, cout, cin, Us.data_ptr(), st))
        def wino():
            check(lib.cp_wino_input_transform_f32(x.data_ptr(), cin, cin, b, h, w, dil, V.data_ptr(), cin, 0, st))
            check(lib.cp_wino_gemm_split_planes_f32(V.data_ptr(), Us.data_ptr(), M.data_ptr(), 36 * tp, tp, cin, cout, 2, st))
            check(lib.cp_wino_output_transform_f32(M.data_ptr(), cout, b, h, w, dil, None, cout, None, None, None, 0, raw.data_ptr(), cout, None, cout, st))
        tw = timeit(wino)
        print("bs %2d %dx%d  %3d -> %3d d%d: direct bf16 %.3f ms = %6.1f TFLOP/s | Winograd hi+mid %.3f ms = %6.1f TFLOP/s-equivalent" % (b, h, w, cin, cout, dil, t, fl / t / 1e9, tw, fl / tw / 1e9))
