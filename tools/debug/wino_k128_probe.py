"""Accuracy of the Winograd path for a K = 128 layer (stage3_unit1_conv1: 128 -> 256, dilation 2) per GEMM route, against fp64:
the whole-network forward error with that layer on / off the Winograd path, and the GEMM alone (cp_wino_gemm_f32 vs split) at K = 128 / 256."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from casapose_amd import _lib
from casapose_amd.engine import split_wino_weights
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
rng = np.random.default_rng(0)
for K, N, T in ((128, 256, 1024), (256, 256, 1024), (128, 128, 1024), (512, 512, 256), (128, 256, 128)):
    V = rng.standard_normal((36, T, K)).astype(np.float32)
    U = (rng.standard_normal((36, N, K)) / np.sqrt(K)).astype(np.float32)
    ref = np.einsum("ptk,pnk->ptn", V.astype(np.float64), U.astype(np.float64))
    Vd, Ud = torch.from_numpy(V).to(dev), torch.from_numpy(U).to(dev)
    M = torch.zeros(36, T, N, device=dev)
    _lib.check(lib.cp_wino_gemm_f32(Vd.data_ptr(), Ud.data_ptr(), M.data_ptr(), 36 * T, T, K, N, st))
    e32 = np.abs(M.cpu().numpy() - ref).max() / np.abs(ref).max()
    Us = split_wino_weights(Ud, 36, N, K)
    out = {}
    for planes in (3, 2):
        M.zero_()
        _lib.check(lib.cp_wino_gemm_split_planes_f32(Vd.data_ptr(), Us.data_ptr(), M.data_ptr(), 36 * T, T, K, N, planes, st))
        out[planes] = np.abs(M.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("K %4d N %4d T %5d: fp32 GEMM %.2e   split x3 %.2e   hi+mid %.2e" % (K, N, T, e32, out[3], out[2]))
