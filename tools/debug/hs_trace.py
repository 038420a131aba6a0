#!/usr/bin/env python3
"""Time line of conv_hsplit's two roles in block 0 (consumer wave 0, loader wave 4), in shader cycles: every barrier (tag 0 arrival, 1 release) and the
stamps between them -- consumers: 2 = slice MFMA loop done, 3 = image block done, 4 = epilogue done; loaders: 2 = register -> LDS stores done, 3 = issues
done, 4 = interpolation done, 5 = twin epilogue done.  Needs the HS_TRACE variant:
    FILES=conv_hsplit bash tools/build_variant.sh HS_TRACE -DHS_TRACE;  CASAPOSE_HIP_LIB=variants/lib_HS_TRACE.so python tools/debug/hs_trace.py [layers]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from casapose_amd import _lib
from casapose_amd.pose_models.tfkeras import Classifiers
dev = torch.device("cuda:0")
net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=9, input_shape=(480, 640, 3), weights=None, device=dev, seed=1, f16x2_guard=False)
img = (2 * torch.rand(16, 480, 640, 3) - 1).to(dev)
net([img])
plan = net._net.plan(16, 480, 640)
raw = C.CDLL(os.environ["CASAPOSE_HIP_LIB"])
raw.cp_hs_trace_read.argtypes = [C.c_void_p]
st = torch.cuda.current_stream(dev).cuda_stream
names = sys.argv[1:] or ["pv_block_4_conv2d", "pv_block_5_conv2d", "pv_block_9_prepare_conv2d", "pv_block_10_prepare_conv2d", "stage1_unit1_conv2", "pv_block_3_conv2d"]
buf = np.zeros((2, 2048), np.uint64)
M56 = np.uint64((1 << 56) - 1)
for c in plan.convs:
    if c.name not in names: continue
    for _ in range(3): c.run(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); c.run(st); e1.record(); torch.cuda.synchronize()
    raw.cp_hs_trace_read(buf.ctypes.data)
    print("== %s  %.3f ms" % (c.name, e0.elapsed_time(e1)))
    for role, rn in ((0, "consumer"), (1, "loader")):
        t = buf[role]
        n = int(t[2047] >> np.uint64(48))
        tags = (t[2:n] >> np.uint64(56)).astype(np.int64)
        tm = (t[2:n] & M56).astype(np.int64)
        real = int(t[2046]) - int(t[0])
        life = int(t[2047] & np.uint64(0xffffffffffff)) - int(t[1] & np.uint64(0xffffffffffff))
        wait = sum(tm[i] - tm[i - 1] for i in range(1, len(tm)) if tags[i] == 1 and tags[i - 1] == 0)
        print("  %-8s wave life %d cycles, %.1f us -> %.2f GHz; waiting at barriers %.0f%% of the traced span" % (rn, life, real / 100.0, life / max(real, 1) / 10.0, 100.0 * wait / max(tm[-1] - tm[0], 1)))
        # a few tiles from the middle of the trace: "tag:+cycles since the previous stamp", a new line at every barrier release
        lo = min(len(tm) // 3, 200)
        rows, cur = [], []
        for i in range(max(lo, 1), len(tm)):
            cur.append("%d:%d" % (tags[i], tm[i] - tm[i - 1]))
            if tags[i] == 1:
                rows.append(" ".join(cur)); cur = []
            if len(rows) >= int(os.environ.get("HS_TRACE_ROWS", "18")): break
        for r in rows: print("      " + r)
