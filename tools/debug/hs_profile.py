#!/usr/bin/env python3
"""Per-section shader-clock time of conv_hsplit's consumer and loader waves for chosen layers of the forward plan.  Needs the HS_PROFILE variant:
    FILES=conv_hsplit bash tools/build_variant.sh HS_PROFILE -DHS_PROFILE;  CASAPOSE_HIP_LIB=variants/lib_HS_PROFILE.so python tools/debug/hs_profile.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casapose_amd import _lib
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=9, input_shape=(480, 640, 3), weights=None, device=dev, seed=1, f16x2_guard=False)
img = (2 * torch.rand(16, 480, 640, 3) - 1).to(dev)
net([img])
plan = net._net.plan(16, 480, 640)
lib = _lib.load()
raw = C.CDLL(os.environ["CASAPOSE_HIP_LIB"])
raw.cp_hs_profile_read.argtypes = [C.c_void_p, C.c_int]
st = torch.cuda.current_stream(dev).cuda_stream
buf = (C.c_ulonglong * 16)()
names = sys.argv[1:] or ["pv_block_4_conv2d", "pv_block_5_conv2d", "pv_block_9_prepare_conv2d", "pv_block_10_prepare_conv2d", "stage1_unit1_conv2", "pv_block_6_prepare_conv2d"]
print("%-30s %8s | consumers: %7s %7s %7s %7s %7s | loaders: %7s %7s %7s %7s  (kilo-cycles per wave per tile; loaders: stores / work, barrier, issues, interpolation)" % ("layer", "tiles/CU", "setup", "mfma", "image", "epi", "barrier", "store", "barrier", "issue", "interp"))
for c in plan.convs:
    if c.name not in names: continue
    c.run(st); torch.cuda.synchronize()
    raw.cp_hs_profile_read(buf, 1)
    reps = 5
    for _ in range(reps): c.run(st)
    torch.cuda.synchronize()
    raw.cp_hs_profile_read(buf, 1)
    d = c.desc
    tiles = ((d.out_h + 7) // 8) * ((d.out_w + 31) // 32) * d.batch * ((d.cout + 63) // 64 if d.cout > 32 else 1)
    per = lambda i: buf[i] / reps / (4.0 * tiles) / 1e3   # 4 waves of a kind per tile
    print("%-30s %8.1f | %18.2f %7.2f %7.2f %7.2f %7.2f | %16.2f %7.2f %7.2f %7.2f" % (c.name, tiles / 256.0, per(0), per(1), per(2), per(3), per(4), per(8), per(9), per(10), per(11)))
