"""One launch shape of cp_conv2d_fwd_bf16_deep, repeated (for rocprofv3 --pmc runs): python tools/debug/bf16_deep_one.py [cin cout dil b h w]"""
import ctypes as C, sys
import numpy as np, torch
sys.path[:0] = ["."]
from casapose_amd import _lib
from casapose_amd._lib import ConvDesc, check
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
a = [int(v) for v in sys.argv[1:]] + [512, 512, 4, 16, 60, 80][len(sys.argv) - 1:]
cin, cout, dil, b, h, w = a
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
for _ in range(5):
    check(lib.cp_conv2d_fwd_bf16_deep(C.byref(d), planes.data_ptr(), st))
torch.cuda.synchronize()
