"""Host-side execution of the CASAPose forward on MI355X.

The arithmetic lives in libcasapose_hip.so (casapose_amd/csrc); this module only owns
device memory (PyTorch-ROCm tensors), folds the inference-mode normalisation layers into
per-channel / per-class affine tables, packs the Keras-layout kernels into the K order
the implicit-GEMM kernel expects, and issues the launches in graph order.

Graph followed: CASAPoseConditional5 (casapose/pose_models/models/pose_models.py:513-635)
over ResNet-18 with output stride 8 (models/resnet.py:183-328).  Layer / weight names are
the Keras names of the reference so weights map 1:1.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, check

BN_EPS = 2e-5  # resnet.py:44; _normalization_layers.py:108

# Hard label maps the forward already computed (arg-max of its own logits), keyed by the storage
# address of the network output: the voting layer receives `torch.split` views of that output and
# can reuse the map instead of re-reading 708 MB of logits.  An entry owns a COPY of the map (the
# plan's own label buffer is overwritten by the next forward), holds a weak reference to the output
# tensor and remembers its version counter: an in-place edit of the output (views share the counter)
# or a dead / re-used storage is a miss, and the voter recomputes the arg-max from the logits.
import weakref

_LABEL_CACHE: Dict[int, Tuple["weakref.ReferenceType", torch.Tensor, int]] = {}


def _remember_labels(out: torch.Tensor, labels: torch.Tensor):
    """keep the label map of `out`; entries whose output tensor has died go first, then the oldest (round 6: a wholesale clear() above eight entries
    could drop the map of an output the caller still holds)"""
    for k in [k for k, (ref, _, _) in _LABEL_CACHE.items() if ref() is None]:
        _LABEL_CACHE.pop(k, None)
    while len(_LABEL_CACHE) >= 8:
        _LABEL_CACHE.pop(next(iter(_LABEL_CACHE)))
    _LABEL_CACHE[out.untyped_storage().data_ptr()] = (weakref.ref(out), labels, out._version)


def cached_labels(storage_ptr: int, shape: Tuple[int, int, int], version: Optional[int] = None) -> Optional[torch.Tensor]:
    hit = _LABEL_CACHE.get(storage_ptr)
    if hit is None:
        return None
    ref, labels, ver = hit
    if ref() is None or tuple(labels.shape) != tuple(shape) or (version is not None and version != ver) or ref()._version != ver:
        _LABEL_CACHE.pop(storage_ptr, None)
        return None
    return labels

STAGE_FILTERS = (64, 128, 256, 512)
STAGE_STRIDE = (1, 2, 1, 1)  # resnet.py:262-290 (output_stride 8)
STAGE_DILATION = (1, 1, 2, 4)
DECODER_DIMS_DEFAULT = (256, 128, 64, 32, 32)
# decoder-2 configuration of blocks 6..10: which use a partial convolution, which upsample their output with the
# label-guided gather (else plain nearest x2).  CASAPoseConditional1-5 (pose_models.py:14-635) differ only in these.
PARTIAL_DEFAULT = (True, True, True, True, True)
GUIDED_DEFAULT = (False, True, True, True, False)
BILINEAR_DEFAULT = (False, False, False, False, False)  # with guided: GuidedBilinearUpsampling (casapose_c_gcu4_bilat)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def fold_bn(params: Dict[str, np.ndarray], name: str, pad_to: Optional[int] = None):
    """Inference-mode (Sync)BatchNormalization as y = x*scale + shift (fp64 fold, fp32 store)."""
    var = params[name + ".moving_variance"].astype(np.float64)
    mean = params[name + ".moving_mean"].astype(np.float64)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    gamma = params.get(name + ".gamma")
    beta = params.get(name + ".beta")
    scale = rstd if gamma is None else gamma.astype(np.float64) * rstd
    shift = -mean * scale
    if beta is not None:
        shift = shift + beta.astype(np.float64)
    if pad_to is not None and pad_to > scale.size:
        scale = np.concatenate([scale, np.zeros(pad_to - scale.size)])
        shift = np.concatenate([shift, np.zeros(pad_to - shift.size)])
    return scale.astype(np.float32), shift.astype(np.float32)


def fold_clade(params: Dict[str, np.ndarray], name: str):
    """ClassAdaptiveWeightedNormalization with a hard label (one-hot mask):
    y = gamma[l]*(x-mean)*rstd + beta[l]  ->  tables [classes][C]
    (_normalization_layers.py:119-139)."""
    var = params[name + ".moving_variance"].astype(np.float64)
    mean = params[name + ".moving_mean"].astype(np.float64)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    gamma = params[name + ".gamma"].astype(np.float64)
    beta = params[name + ".beta"].astype(np.float64)
    scale = gamma * rstd[None, :]
    shift = beta - gamma * (mean * rstd)[None, :]
    return scale.astype(np.float32), shift.astype(np.float32)


class FusedConv:
    """One cp_conv2d_fwd_f32 launch with its packed weights and descriptor."""

    def __init__(self, name: str, kernel: np.ndarray, layout: int, kh: int, kw: int, cout: int,
                 sources: Sequence[Tuple[int, int]], device: torch.device, want_split: bool = True):
        lib = _lib.load()
        self.name = name
        self.kh, self.kw, self.cout = kh, kw, cout
        self.sources = list(sources)
        ns = len(sources)
        chans = (C.c_int * 2)(*([s[0] for s in sources] + [0] * (2 - ns)))
        real = (C.c_int * 2)(*([s[1] for s in sources] + [0] * (2 - ns)))
        self.ktot = lib.cp_conv_ktot(kh, kw, ns, chans)
        w = np.ascontiguousarray(kernel, dtype=np.float32)
        cin = sum(s[1] for s in sources)
        expect = (kh, kw, cin, cout) if layout == 0 else (cin, kh, kw, cout)
        if tuple(w.shape) != expect:
            raise ValueError("%s: kernel shape %s, expected %s" % (name, w.shape, expect))
        self.wp = None
        self.wp_stem_split_f32 = None
        self.stem_split = 0     # 3 / 1: this binding runs the stem on the bf16 matrix pipe (set by ForwardPlan)
        if all(s[0] == 4 or s[0] % 32 == 0 for s in sources):   # (16-multiple sources exist only for the bf16-pipe kernel below)
            packed = np.empty((cout, self.ktot), dtype=np.float32)
            check(lib.cp_conv_pack_weights_host(w.ctypes.data, layout, kh, kw, cout, ns, chans, real, packed.ctypes.data),
                  "cp_conv_pack_weights_host(%s)" % name)
            self.wp = torch.from_numpy(packed).to(device)
        # second packing for the LDS-resident halo-tile kernel (3x3, cout <= 64, 32-multiple sources [+ image])
        self.wp_halo = None
        halo_ok = kh == 3 and kw == 3 and cout <= 64 and sources[0][0] % 32 == 0 and (ns == 1 or sources[1][0] == 4 or sources[1][0] % 32 == 0)
        if halo_ok:
            nfl = lib.cp_conv_halo_weight_floats(cout, ns, chans)
            ph = np.empty(nfl, dtype=np.float32)
            check(lib.cp_conv_pack_weights_halo_host(w.ctypes.data, layout, cout, ns, chans, real, ph.ctypes.data),
                  "cp_conv_pack_weights_halo_host(%s)" % name)
            self.wp_halo = torch.from_numpy(ph).to(device)
        if kh == 7 and kw == 7 and cout == 64 and ns == 1 and sources[0][0] == 4:  # the ResNet stem: packing of csrc/conv_stem.hip (used at stride 2 / pad 3)
            ph = np.empty(25 * 2 * 64 * 4, dtype=np.float32)
            check(lib.cp_conv_pack_weights_stem_host(w.ctypes.data, layout, sources[0][1], ph.ctypes.data), "cp_conv_pack_weights_stem_host(%s)" % name)
            self.wp_halo = torch.from_numpy(ph).to(device)
            if want_split:   # the same layer on the bf16 matrix pipe (csrc/conv_stem_split.hip): fp32 image of its fragment stream
                ps = np.empty(lib.cp_conv_stem_split_weight_floats(), dtype=np.float32)
                check(lib.cp_conv_pack_weights_stem_split_host(w.ctypes.data, layout, sources[0][1], ps.ctypes.data), "cp_conv_pack_weights_stem_split_host(%s)" % name)
                self.wp_stem_split_f32 = torch.from_numpy(ps).to(device)
        # third packing: the fp32 image of the fragment stream of csrc/conv_hsplit.hip (3x3 / cout <= 512 in passes of 64 / 16-multiple sources [+ image]);
        # its bf16 planes (3 = exact split, 1 = plain bf16) are made on the device when a mode asks for them
        self.wp_split_f32 = None
        self._split_planes: Dict[int, torch.Tensor] = {}
        self._descale: Dict[int, float] = {}
        self.split_mode = 0
        split_ok = (kh == 3 and kw == 3 and cout <= 512 and cout % 4 == 0 and sources[0][0] % 16 == 0 and sources[0][0] != 4
                    and (ns == 1 or (sources[1][0] == 4 and cout <= 32) or sources[1][0] % 16 == 0))
        if split_ok and want_split:
            nfl = lib.cp_conv_split_weight_floats(cout, ns, chans)
            ps = np.empty(nfl, dtype=np.float32)
            check(lib.cp_conv_pack_weights_split_host(w.ctypes.data, layout, cout, ns, chans, real, ps.ctypes.data), "cp_conv_pack_weights_split_host(%s)" % name)
            self.wp_split_f32 = torch.from_numpy(ps).to(device)
        self.desc = ConvDesc()
        self._keep: List[torch.Tensor] = []
        self.head_w: Optional[torch.Tensor] = None
        self.head_cout = 0
        self._gemm: Optional[dict] = None   # set by enable_gemm_split() for the CURRENT binding only
        self.head_in_scale, self._head_tabs = 1.0, None   # f16x2 range guard: power of two on the fused head's operand (set_head_in_scale)
        self.deep_bf16 = False              # bf16 conv mode: this binding runs on csrc/conv_bf16d.hip (set by ForwardPlan)
        self.mon_ptr: Optional[int] = None  # f16x2 range monitor: device address of this layer's slot while a forward runs armed (ForwardPlan._run_armed)

    def _make_planes(self, key: int, image: torch.Tensor, planes: int, stream: Optional[int]) -> torch.Tensor:
        """2-byte operand planes of an fp32 fragment image: 3 = exact bf16 split, 1 = bf16, PLANES_F16X2 = the fp16 two-way split of image * 2^e (the
        inverse factor is kept in _descale[key]: the kernels multiply their accumulators by it)."""
        if key not in self._split_planes:
            lib = _lib.load()
            out = torch.empty(image.numel() // 512 * (planes & 15) * 1024, dtype=torch.uint8, device=image.device)
            st = torch.cuda.current_stream(out.device).cuda_stream if stream is None else stream
            scale = float(lib.cp_f16x2_weight_scale(float(image.abs().max()))) if planes == _lib.PLANES_F16X2 else 1.0
            check(lib.cp_conv_split_weights_scaled_f32(image.data_ptr(), image.numel(), planes, scale, out.data_ptr(), st), "cp_conv_split_weights_scaled_f32")
            self._split_planes[key] = out
            self._descale[key] = 1.0 / scale
        return self._split_planes[key]

    def split_weights(self, planes: int, stream: Optional[int] = None) -> torch.Tensor:
        """operand planes of the weights for cp_conv2d_fwd_split / cp_conv2d_fwd_stem_split (made once per mode)."""
        image = self.wp_split_f32 if self.wp_split_f32 is not None else self.wp_stem_split_f32
        return self._make_planes(planes, image, planes, stream)

    def head_split_weights(self, planes: int, stream: Optional[int] = None) -> torch.Tensor:
        return self._make_planes(-planes, self.head_w_split_f32, planes, stream)

    def attach_head(self, kernel_1x1: np.ndarray):
        """Fuse a following 1x1 convolution (HWIO [1,1,32,q], no bias / activation) into this layer's epilogue."""
        lib = _lib.load()
        w = np.ascontiguousarray(kernel_1x1, dtype=np.float32).reshape(32, -1)
        if self.cout != 32 or self.wp_halo is None or not (1 <= w.shape[1] <= 32):
            raise ValueError("%s: a fused head needs a 3x3 halo-kernel layer with 32 output channels" % self.name)
        packed = np.empty(1024, dtype=np.float32)
        check(lib.cp_conv_pack_head_weights_host(w.ctypes.data, w.shape[1], packed.ctypes.data), "cp_conv_pack_head_weights_host")
        self.head_w = torch.from_numpy(packed).to(self.wp_halo.device)
        self.head_cout = int(w.shape[1])
        ps = np.empty(1024, dtype=np.float32)   # the same head for the bf16-pipe kernel
        check(lib.cp_conv_pack_head_split_host(w.ctypes.data, w.shape[1], ps.ctypes.data), "cp_conv_pack_head_split_host")
        self.head_w_split_f32 = torch.from_numpy(ps).to(self.wp_halo.device)

    def bind(self, *, batch, in_h, in_w, stride=1, dilation=1, pad=0, srcs, tap_label=None, row_scale=None,
             residual=None, scale=None, shift=None, epi_label=None, act=0, out_raw=None, out_raw_ld=None,
             out_act=None, out_act_ld=None, tile_hint=0, head_out=None, head_out_ld=0, head_label_out=None, head_label_classes=0):
        """srcs: list of dicts(data=tensor, ld=int, mode=int, sel=tensor|None, pre=(scale,shift)|None)."""
        d = self.desc
        self._gemm = None   # a GEMM route decided for an earlier shape must not survive a re-bind (rows would be stale: out-of-bounds launch)
        self.deep_bf16 = False
        self.head_in_scale, self._head_tabs, self._epi_tabs = 1.0, None, (scale, shift)
        eh = (self.kh - 1) * dilation + 1
        ew = (self.kw - 1) * dilation + 1
        d.batch, d.in_h, d.in_w = batch, in_h, in_w
        d.out_h = (in_h + 2 * pad - eh) // stride + 1
        d.out_w = (in_w + 2 * pad - ew) // stride + 1
        d.cout, d.kh, d.kw = self.cout, self.kh, self.kw
        d.stride, d.dilation, d.pad = stride, dilation, pad
        d.num_sources = len(srcs)
        self._srcs = list(srcs)   # (ForwardPlan.f16x2_operand_ranges reads the source tensors back)
        keep = [self.wp] if self.wp is not None else []
        for i, s in enumerate(srcs):
            cs = d.src[i]
            cs.data = _ptr(s["data"])
            cs.channels = self.sources[i][0]
            cs.ld = s["ld"]
            cs.mode = s.get("mode", _lib.SRC_DIRECT)
            cs.sel = _ptr(s.get("sel"))
            pre = s.get("pre")
            cs.pre_scale = _ptr(pre[0]) if pre else None
            cs.pre_shift = _ptr(pre[1]) if pre else None
            keep += [s["data"], s.get("sel")] + (list(pre) if pre else [])
        d.weights = _ptr(self.wp)
        d.weights_halo = _ptr(self.wp_halo)
        d.tap_label = _ptr(tap_label)
        d.row_scale = _ptr(row_scale)
        d.residual = _ptr(residual)
        d.residual_ld = self.cout
        d.scale, d.shift = _ptr(scale), _ptr(shift)
        d.epi_label = _ptr(epi_label)
        d.act = act
        d.out_raw = _ptr(out_raw)
        d.out_raw_ld = out_raw_ld if out_raw_ld is not None else self.cout
        d.out_act = _ptr(out_act)
        d.out_act_ld = out_act_ld if out_act_ld is not None else self.cout
        # tile_hint TILE_SPLIT3 / TILE_BF16 / TILE_F16X2 (Python-side values): run on csrc/conv_hsplit.hip when the layer is in its range
        self.split_mode = 0
        if tile_hint in (_lib.TILE_SPLIT3, _lib.TILE_BF16, _lib.TILE_F16X2):
            if self.wp_split_f32 is None:
                raise _lib.CasaposeHipError("%s: the bf16-pipe kernel covers 3x3 / cout <= 512 / 16-multiple sources only" % self.name)
            self.split_mode = {_lib.TILE_SPLIT3: 3, _lib.TILE_BF16: 1, _lib.TILE_F16X2: _lib.PLANES_F16X2}[tile_hint]
            tile_hint = 0
        d.tile_hint = tile_hint
        d.head_weights = _ptr(self.head_w) if head_out is not None else None
        d.head_out = _ptr(head_out)
        d.head_cout = self.head_cout if head_out is not None else 0
        d.head_out_ld = head_out_ld
        d.head_label_out = _ptr(head_label_out) if head_out is not None else None
        d.head_label_classes = head_label_classes if head_label_out is not None else 0
        keep += [head_label_out, tap_label, row_scale, residual, scale, shift, epi_label, out_raw, out_act]
        self._keep = [k for k in keep if k is not None]
        return d.out_h, d.out_w

    def enable_gemm_split(self, kernel_hwio: np.ndarray, planes: int) -> bool:
        """1x1 / stride 1 / one source / plain output: the layer is a GEMM rows x cin x cout, which the bf16-pipe GEMM of the Winograd path
        (csrc/wino_gemm_split.hip) computes as it is -- 3 planes exact split (fp32-equivalent), 2 planes hi + mid (the bf16 conv mode).
        Called by the forward plan in the opt-in conv modes after bind(); False when the descriptor is outside that shape."""
        d = self.desc
        rows = d.batch * d.out_h * d.out_w
        cin = self.sources[0][0]
        ok = (self.kh == 1 and self.kw == 1 and d.stride == 1 and d.pad == 0 and d.num_sources == 1 and d.src[0].mode == _lib.SRC_DIRECT
              and not d.src[0].pre_scale and d.src[0].ld == cin and self.sources[0][1] == cin and cin % 32 == 0 and rows % 128 == 0
              and d.out_raw and d.out_raw_ld == self.cout and not d.out_act and not d.scale and not d.residual and not d.row_scale
              and not d.tap_label and not d.head_out and d.act == 0)
        self._gemm = None
        if not ok:
            return False
        U = torch.from_numpy(np.ascontiguousarray(kernel_hwio.reshape(cin, self.cout).T[None])).to(torch.float32).to(self.wp.device if self.wp is not None else "cuda")
        if planes == _lib.PLANES_F16X2:
            Us, c_scale = split_wino_weights_f16x2(U, 1, self.cout, cin)
        else:
            Us, c_scale = split_wino_weights(U, 1, self.cout, cin), 1.0
        self._gemm = dict(Us=Us, rows=rows, k=cin, planes=planes, c_scale=c_scale)
        return True

    def set_head_in_scale(self, factor: float):
        """The fused 1x1 head of an f16x2 binding converts this layer's activated output t = act(v * scale + shift), which exists in registers only.
        Multiply the tables by a power of two: act(2^e x) = 2^e act(x) exactly for ReLU / leaky ReLU, so the head sees 2^e t and its accumulator
        factor (head_descale) takes 2^-e.  Only where nothing else receives t (no out_act) and a table exists."""
        d = self.desc
        scale, shift = self._epi_tabs
        if not d.head_out or d.out_act or scale is None or self.split_mode != _lib.PLANES_F16X2:
            raise _lib.CasaposeHipError("%s: no fused f16x2 head whose operand could be scaled" % self.name)
        self.head_in_scale = float(factor)
        self._head_tabs = (scale * factor, shift * factor)
        d.scale, d.shift = self._head_tabs[0].data_ptr(), self._head_tabs[1].data_ptr()

    def f16x2_active(self) -> bool:
        """does the CURRENT binding convert operands with the fp16 two-way split?"""
        return _lib.PLANES_F16X2 in (self.split_mode, self.stem_split, (self._gemm or {}).get("planes", 0))

    def demote_to_exact_split(self, kernel_hwio: Optional[np.ndarray] = None):
        """this binding's f16x2 launches on the exact three-way bf16 split instead (planes are made on first use)"""
        if self.split_mode == _lib.PLANES_F16X2:
            self.split_mode = 3
        if self.stem_split == _lib.PLANES_F16X2:
            self.stem_split = 3
        if self._gemm is not None and self._gemm["planes"] == _lib.PLANES_F16X2:
            if kernel_hwio is None or not self.enable_gemm_split(kernel_hwio, 3):
                raise _lib.CasaposeHipError("%s: cannot re-route the 1x1 GEMM to the exact split" % self.name)

    def run(self, stream: int):
        """One launch; while the plan runs armed (mon_ptr set) an f16x2 binding reports max |x| of what it converts into its monitor slot."""
        if self.mon_ptr and self.f16x2_active():
            lib = _lib.load()
            lib.cp_f16x2_monitor_set(self.mon_ptr)
            try:
                self._launch(stream)
            finally:
                lib.cp_f16x2_monitor_set(None)
            return
        self._launch(stream)

    def _launch(self, stream: int):
        lib = _lib.load()
        g = self._gemm
        if g is not None:
            d = self.desc
            assert g["rows"] == d.batch * d.out_h * d.out_w, "%s: GEMM route bound for another shape" % self.name
            check(lib.cp_wino_gemm_split_scaled_f32(d.src[0].data, g["Us"].data_ptr(), d.out_raw, g["rows"], g["rows"], g["k"], self.cout, g["planes"], g["c_scale"],
                                                    stream), "cp_wino_gemm_split_scaled_f32(%s)" % self.name)
            return
        if self.deep_bf16:
            check(lib.cp_conv2d_fwd_bf16_deep(C.byref(self.desc), self.split_weights(1, stream).data_ptr(), stream), "cp_conv2d_fwd_bf16_deep(%s)" % self.name)
            return
        if self.stem_split:
            wsp = self.split_weights(self.stem_split, stream)
            check(lib.cp_conv2d_fwd_stem_split_scaled(C.byref(self.desc), wsp.data_ptr(), self.stem_split, self._descale[self.stem_split], stream),
                  "cp_conv2d_fwd_stem_split(%s)" % self.name)
            return
        if self.split_mode:
            if not lib.cp_conv_split_applicable(C.byref(self.desc)):
                raise _lib.CasaposeHipError("%s: descriptor outside the range of cp_conv2d_fwd_split" % self.name)
            head = self.head_split_weights(self.split_mode, stream).data_ptr() if self.desc.head_out else None
            wsp = self.split_weights(self.split_mode, stream)
            check(lib.cp_conv2d_fwd_split_scaled(C.byref(self.desc), wsp.data_ptr(), head, self.split_mode, self._descale[self.split_mode],
                                                 self._descale[-self.split_mode] / self.head_in_scale if head else 1.0, stream), "cp_conv2d_fwd_split(%s)" % self.name)
            return
        check(lib.cp_conv2d_fwd_f32(C.byref(self.desc), stream), "cp_conv2d_fwd_f32(%s)" % self.name)

    @property
    def flops(self) -> float:
        d = self.desc
        cin = sum(s[1] for s in self.sources)
        head = 2.0 * d.batch * d.out_h * d.out_w * 32 * d.head_cout if d.head_out else 0.0
        return 2.0 * d.batch * d.out_h * d.out_w * self.kh * self.kw * cin * self.cout + head


WINO_GROUPED_CONV = os.environ.get("CASAPOSE_WINO_GROUPED_CONV", "0") == "1"
# The Winograd GEMM as exact 3-way bf16 splits on the bf16 matrix pipe (fp32-EQUIVALENT: every product exact, fp32 accumulation, measured
# error <= the fp32 MFMA's; csrc/wino_gemm_split.hip).  Round 3: the DEFAULT of both plans -- the training plan since round 2, the
# inference plan whenever its conv mode is "split" (its default, see CasaposeNet); CASAPOSE_WINO_GEMM=f32 restores the fp32 MFMA in both,
# CASAPOSE_WINO_GEMM=split forces the split GEMM even with conv_mode="f32".
WINO_GEMM_SPLIT = os.environ.get("CASAPOSE_WINO_GEMM", "") == "split"
WINO_GEMM_F32 = os.environ.get("CASAPOSE_WINO_GEMM", "") == "f32"
TRAIN_WINO_GEMM_SPLIT = os.environ.get("CASAPOSE_WINO_GEMM", "split") == "split"
# default arithmetic of the inference plan's convolutions (CasaposeNet(conv_mode=None)):
#   "f16x2" (default since round 4) = every fp32 operand as an fp16 pair hi + lo (reproduced to within one fp32 ulp), the three exact products hi*hi, hi*lo, lo*hi
#           accumulated in fp32 on the fp16 matrix pipe, for every layer the split kernels cover: fp32-LEVEL -- against fp64 its error is at or below
#           the fp32 MFMA's on the GEMM, on single convolutions and on the whole network (tests/test_gpu_f16x2.py) -- with half the MFMAs of "split";
#   "split" = exact three-way bf16 splits, six products (fp32-equivalent for any finite operand, no range conditions: the round-3 default);
#   "f32"   = the fp32 MFMA everywhere;   "bf16" = bf16 operands (3e-2 gates)
DEFAULT_INFER_CONV_MODE = "f16x2"
# The range condition of f16x2 (csrc/split_f16.h) is CHECKED, per layer, on the first forward of every plan (and again after set_params / load_weights,
# which drop the plans): a layer whose converted operands leave [F16X2_AMAX_LO, F16X2_AMAX_HI] gets an exact remedy -- a power of two on a Winograd layer's
# V or on a fused head's operand, the exact three-way bf16 split for a direct layer (ForwardPlan._calibrate) -- with one warning naming the layers.  Below LO the low halves are fp16 subnormals (absolute 2^-25: worse than 2^-24 of the tensor's
# maximum); HI = 65504 / 4 leaves two octaves for later batches before a conversion clamps (and a clamp is graceful, split_f16.h: the maximum over a
# batch of 4800 tiles x 36 planes x K channels moves by tens of per cent between batches, not by factors).  CASAPOSE_F16X2_GUARD=0 / CasaposeNet(f16x2_guard=False)
# switch the check off (the unguarded plan of round 4: tests compare the two).
F16X2_GUARD = os.environ.get("CASAPOSE_F16X2_GUARD", "1") != "0"
F16X2_AMAX_LO = 0.5
F16X2_AMAX_HI = 65504.0 / 4.0
# Round 6: the measurements come from the converting kernels themselves (monitor slots behind the C ABI: cp_f16x2_monitor_set, include/casapose_hip.h) and
# the guard STAYS on after the calibration: EVERY forward of a guarded plan runs armed, so the slots are sticky maxima over everything the plan has
# converted since they were last read (measured cost, A/B in one call: 2149.5 -> 2141.5 images/s, 0.37 %; CASAPOSE_F16X2_MONITOR=0 arms the calibration
# passes only).  Every F16X2_MONITOR_EVERY-th forward the slots are copied to pinned host memory asynchronously, zeroed behind the copy, and judged at
# the start of a later forward -- no synchronisation on the hot path.  A layer whose converted maximum has left [LO / SLACK, HI * SLACK] (a later
# batch far above or below the one the plan was calibrated on) re-arms the calibration with one warning.  SLACK = 2 keeps HI * SLACK at half the
# fp16 maximum: nothing clamps before the monitor reacts.
F16X2_MONITOR = os.environ.get("CASAPOSE_F16X2_MONITOR", "1") != "0"
F16X2_MONITOR_EVERY = max(1, int(os.environ.get("CASAPOSE_F16X2_MONITOR_EVERY", "8")))
F16X2_MONITOR_SLACK = 2.0
# the last fused head writes whole output records (ForwardPlan._whole_records).  Opt-in: three one-call A/Bs on three boxes gave +0.7 %, -0.1 % and -0.2 %
# on the step (block 10's epilogue pays 0.05-0.06 ms for the copy; what comes back from the kernels that no longer share HBM with read-modify-writes
# depends on the box) -- not a gain one can rely on.  The training plan's heads, stand-alone streaming kernels, do write whole records by default.
WHOLE_RECORDS = os.environ.get("CASAPOSE_INFER_HEAD_RECORDS", "0") == "1"
BF16_DEEP = os.environ.get("CASAPOSE_BF16_DEEP", "1") != "0"   # bf16 conv mode: deep layers on csrc/conv_bf16d.hip (0: two-plane Winograd)
# images per Winograd batch group (0 = the whole batch in one go)
WINO_CHUNK = int(os.environ.get("CASAPOSE_WINO_CHUNK", "0"))
MATERIALISE_BILINEAR = os.environ.get("CASAPOSE_MATERIALISE_BILINEAR", "0") == "1"
WINO_FUSE_OUT_IN = os.environ.get("CASAPOSE_WINO_FUSE_OUT_IN", "1") != "0"
STEM_SPLIT = os.environ.get("CASAPOSE_STEM_SPLIT", "1") != "0"   # A/B switch: conv0 on csrc/conv_stem_split.hip in the split / bf16 conv modes
# two-stream forward (CasaposeNet._forward_two_streams): half-batches pipelined over a matrix-pipe stream and an HBM stream
TWO_STREAM = os.environ.get("CASAPOSE_TWO_STREAM", "0") == "1"
TWO_STREAM_BLOCKS = int(os.environ.get("CASAPOSE_TWO_STREAM_BLOCKS", "224"))   # blocks of the persistent kernels while both streams run
TWO_STREAM_MODE = os.environ.get("CASAPOSE_TWO_STREAM_MODE", "half")   # "half": a stream per half-batch; "tag": a stream per kernel class
TWO_STREAM_SKEW = float(os.environ.get("CASAPOSE_TWO_STREAM_SKEW", "0.4"))     # fraction of its steps the first half runs ahead   # A/B switch of the fused output -> input transform


def split_wino_weights(U: torch.Tensor, groups: int, n: int, k: int, out: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
    """cp_wino_split_weights_f32: [groups][n][k] fp32 -> the fragment-major hi / mid / lo bf16 planes cp_wino_gemm_split_f32 reads."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(lib.cp_wino_split_weights_bytes(groups, n, k), dtype=torch.uint8, device=U.device)
    st = torch.cuda.current_stream(U.device).cuda_stream if stream is None else stream
    check(lib.cp_wino_split_weights_f32(U.data_ptr(), groups, n, k, out.data_ptr(), st), "cp_wino_split_weights_f32")
    return out


def split_wino_weights_f16x2(U: torch.Tensor, groups: int, n: int, k: int, stream: Optional[int] = None):
    """cp_wino_split_weights_scaled_f32 with CP_PLANES_F16X2: the fp16 two-way split of U * 2^e (e from max |U|: csrc/split_f16.h).  Returns the
    plane buffer and c_scale = 2^-e, the factor cp_wino_gemm_split_scaled_f32 multiplies its accumulators by."""
    lib = _lib.load()
    out = torch.empty(lib.cp_wino_split_weights_bytes(groups, n, k), dtype=torch.uint8, device=U.device)
    st = torch.cuda.current_stream(U.device).cuda_stream if stream is None else stream
    scale = float(lib.cp_f16x2_weight_scale(float(U.abs().max())))
    check(lib.cp_wino_split_weights_scaled_f32(U.data_ptr(), groups, n, k, _lib.PLANES_F16X2, scale, out.data_ptr(), st), "cp_wino_split_weights_scaled_f32")
    return out, 1.0 / scale


class WinoConv:
    """A deep 3x3 / stride-1 convolution executed as Winograd F(4x4,3x3): input transform(s), ONE grouped 1x1 launch over
    the 36 planes, output transform with the fused epilogue (csrc/wino.hip).  Same interface as FusedConv.run()."""

    def __init__(self, name: str, kernel_hwio: np.ndarray, cout: int, sources: Sequence[Tuple[int, int]], device: torch.device,
                 split_planes: Optional[int] = None):
        lib = _lib.load()
        self.name, self.cout, self.sources = name, cout, list(sources)
        self.kh = self.kw = 3
        self.ktot = sum(s[0] for s in sources)
        if self.ktot % 32 or cout % 4:
            raise ValueError("%s: Winograd path needs 32-multiple input channels and 4-multiple output channels" % name)
        w = np.ascontiguousarray(kernel_hwio, dtype=np.float32)
        cin = sum(s[1] for s in sources)
        if tuple(w.shape) != (3, 3, cin, cout):
            raise ValueError("%s: kernel shape %s, expected %s" % (name, w.shape, (3, 3, cin, cout)))
        U = np.zeros((36, cout, self.ktot), np.float32)
        c0 = k0 = 0
        for cpad, creal in sources:
            check(lib.cp_wino_pack_weights_host(w.ctypes.data, cin, cout, c0, cpad, creal, self.ktot, k0, U.ctypes.data), "cp_wino_pack_weights_host(%s)" % name)
            c0 += creal
            k0 += cpad
        self.U = torch.from_numpy(U).to(device)
        # pre-split bf16 planes of the opt-in GEMM; split_planes = 2 (the bf16 conv mode): hi + mid planes only, three products, NOT fp32-equivalent
        self.planes = split_planes or 3
        self.c_scale = 1.0
        if self.planes == _lib.PLANES_F16X2:   # the fp16 two-way split: weights times a power of two, the GEMM undoes it
            self.Us, self.c_scale = split_wino_weights_f16x2(self.U, 36, cout, self.ktot)
        else:
            self.Us = split_wino_weights(self.U, 36, cout, self.ktot) if (WINO_GEMM_SPLIT or split_planes) else None
        self.desc = ConvDesc()  # the grouped GEMM
        self._keep: List = []
        self.v_scale, self._vs_vec, self._next_tabs = 1.0, None, None
        self.fuse_next, self.skip_input = None, False
        self.mon_ptr: Optional[int] = None   # f16x2 range monitor slot of THIS layer's V while a forward runs armed

    def demote_to_exact_split(self):
        """the fp16 two-way split of this layer's GEMM replaced by the exact three-way bf16 split (operands outside the fp16 range condition)"""
        if self.planes == _lib.PLANES_F16X2:
            self.Us, self.c_scale, self.planes = split_wino_weights(self.U, 36, self.cout, self.ktot), 1.0, 3

    @staticmethod
    def tiles(batch, h, w, dilation) -> Tuple[int, int]:
        t, tp = C.c_int(0), C.c_int(0)
        check(_lib.load().cp_wino_tiles(batch, h, w, dilation, C.byref(t), C.byref(tp)), "cp_wino_tiles")
        return t.value, tp.value

    def bind(self, *, batch, in_h, in_w, dilation, srcs, V, M, residual=None, scale=None, shift=None, epi_label=None, act=0,
             out_raw=None, out_act=None):
        """srcs: list of dicts(data=tensor, ld=int); V / M: scratch tensors of at least 36*Tp*ktot and 36*Tp*cout floats."""
        self.batch, self.h, self.w, self.dil = batch, in_h, in_w, dilation
        self.T, self.Tp = self.tiles(batch, in_h, in_w, dilation)
        if V.numel() < 36 * self.Tp * self.ktot or M.numel() < 36 * self.Tp * self.cout:
            raise ValueError("%s: Winograd scratch too small" % self.name)
        self.srcs, self.V, self.M = list(srcs), V, M
        self.v_scale, self._vs_vec, self._next_tabs = 1.0, None, None   # f16x2 range guard (ForwardPlan._calibrate): power of two applied to V
        self.epi = dict(residual=residual, scale=scale, shift=shift, epi_label=epi_label, act=act, out_raw=out_raw, out_act=out_act)
        d = self.desc
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w = 36, 1, self.Tp, 1, self.Tp
        d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = self.cout, 1, 1, 1, 1, 0
        d.num_sources = 1
        d.src[0].data, d.src[0].channels, d.src[0].ld, d.src[0].mode = V.data_ptr(), self.ktot, self.ktot, _lib.SRC_DIRECT
        d.src[0].sel = d.src[0].pre_scale = d.src[0].pre_shift = None
        d.weights, d.weights_halo = self.U.data_ptr(), None
        d.tap_label = d.row_scale = d.residual = d.scale = d.shift = d.epi_label = None
        d.residual_ld, d.act = self.cout, 0
        d.out_raw, d.out_raw_ld, d.out_act, d.out_act_ld = M.data_ptr(), self.cout, None, self.cout
        d.tile_hint = 0
        d.head_weights = d.head_out = None
        d.head_label_out, d.head_label_classes = None, 0
        d.head_cout = d.head_out_ld = 0
        d.group_rows, d.group_weight_stride = self.Tp, self.cout * self.ktot
        self._keep = [V, M, residual, scale, shift, epi_label, out_raw, out_act] + [s["data"] for s in srcs]
        return in_h, in_w

    class _Armed:
        """context: the transform launched inside reports max |V| of what it writes into the monitor slot at `ptr` (None: nothing armed)"""
        __slots__ = ("ptr",)

        def __init__(self, ptr):
            self.ptr = None if WINO_GROUPED_CONV else ptr   # (the grouped mode of the general kernel converts nothing to fp16)

        def __enter__(self):
            if self.ptr:
                _lib.load().cp_f16x2_monitor_set(self.ptr)

        def __exit__(self, *exc):
            if self.ptr:
                _lib.load().cp_f16x2_monitor_set(None)
            return False

    def _armed(self, ptr: Optional[int]):
        return WinoConv._Armed(ptr)

    def _input_transform(self, src_ptr: int, ld: int, cpad: int, nb: int, off: int, stream: int):
        """V[.., off : off + cpad] = B^T d B of one source; with v_scale != 1 through the transform's per-channel input affine (x * 2^e + 0: exact)"""
        with self._armed(self.mon_ptr if self.planes == _lib.PLANES_F16X2 else None):
            self._input_transform_launch(src_ptr, ld, cpad, nb, off, stream)

    def _input_transform_launch(self, src_ptr: int, ld: int, cpad: int, nb: int, off: int, stream: int):
        lib = _lib.load()
        if self.v_scale == 1.0:
            check(lib.cp_wino_input_transform_f32(src_ptr, ld, cpad, nb, self.h, self.w, self.dil, self.V.data_ptr(), self.ktot, off, stream),
                  "cp_wino_input_transform_f32(%s)" % self.name)
            return
        if self._vs_vec is None or self._vs_vec[0] != self.v_scale:
            n = max(c for c, _ in self.sources)
            self._vs_vec = (self.v_scale, torch.full((n,), self.v_scale, dtype=torch.float32, device=self.V.device), torch.zeros(n, dtype=torch.float32, device=self.V.device))
        check(lib.cp_wino_input_transform_pre_f32(src_ptr, ld, cpad, nb, self.h, self.w, self.dil, self.V.data_ptr(), self.ktot, off,
                                                  self._vs_vec[1].data_ptr(), self._vs_vec[2].data_ptr(), _lib.ACT_NONE, stream), "cp_wino_input_transform_pre_f32(%s)" % self.name)

    def _tables_for_next(self):
        """(scale, shift) pointers of this layer's epilogue when its fused output -> input transform writes the NEXT layer's V: the tables times the
        next layer's v_scale (ReLU / leaky ReLU commute with a positive factor, a power of two is exact) -- the activated map itself is not stored"""
        e, nxt = self.epi, self.fuse_next
        if nxt.v_scale == 1.0:
            return _ptr(e["scale"]), _ptr(e["shift"])
        if e["scale"] is None:
            raise _lib.CasaposeHipError("%s: cannot scale the fused transform's output without a normalisation table" % self.name)
        if self._next_tabs is None or self._next_tabs[0] != nxt.v_scale:
            self._next_tabs = (nxt.v_scale, e["scale"] * nxt.v_scale, e["shift"] * nxt.v_scale)
        return self._next_tabs[1].data_ptr(), self._next_tabs[2].data_ptr()

    def chunks(self) -> List[Tuple[int, int, int]]:
        """[(first image, images, padded tiles)]: the batch is processed in groups of WINO_CHUNK images so that the two scratch tensors
        of a group (V: 36*Tp*K, M: 36*Tp*Cout floats -- 2.25x the layer's input and output) stay resident in the 256 MiB Infinity Cache
        between the input transform, the GEMM and the output transform instead of making two round trips to HBM each."""
        # the grouped mode of the general kernel reads desc.group_rows = the whole batch's Tp: chunking does not apply to it
        n = WINO_CHUNK if 0 < WINO_CHUNK < self.batch and not WINO_GROUPED_CONV else self.batch
        return [(b0, min(n, self.batch - b0), self.tiles(min(n, self.batch - b0), self.h, self.w, self.dil)[1]) for b0 in range(0, self.batch, n)]

    def run(self, stream: int):
        lib = _lib.load()
        e = self.epi
        px = self.h * self.w
        at = lambda t, b0, ld, size=4: (t.data_ptr() + b0 * px * ld * size) if t is not None else None  # noqa: E731
        for b0, nb, tp in self.chunks():
            off = 0
            for (cpad, _), s in zip(self.sources, self.srcs):
                if getattr(self, "skip_input", False):   # the producer's fused output -> input transform has written V already
                    break
                self._input_transform(at(s["data"], b0, s["ld"]), s["ld"], cpad, nb, off, stream)
                off += cpad
            self.run_gemm(stream, tp)
            nxt = getattr(self, "fuse_next", None)
            if nxt is not None:   # Y = A^T M A + epilogue, then straight into the next layer's V (the activated map stays on chip)
                sc_, sh_ = self._tables_for_next()
                with self._armed(nxt.mon_ptr if nxt.planes == _lib.PLANES_F16X2 else None):   # it writes the NEXT layer's V
                    check(lib.cp_wino_output_input_transform_f32(self.M.data_ptr(), self.cout, nb, self.h, self.w, self.dil, at(e["residual"], b0, self.cout), self.cout,
                                                                 sc_, sh_, e["act"], at(e["out_raw"], b0, self.cout), self.cout, None, self.cout,
                                                                 self.V.data_ptr(), nxt.ktot, 0, stream), "cp_wino_output_input_transform_f32(%s)" % self.name)
                continue
            check(lib.cp_wino_output_transform_f32(self.M.data_ptr(), self.cout, nb, self.h, self.w, self.dil, at(e["residual"], b0, self.cout), self.cout,
                                                   _ptr(e["scale"]), _ptr(e["shift"]), at(e["epi_label"], b0, 1, 1), e["act"], at(e["out_raw"], b0, self.cout),
                                                   self.cout, at(e["out_act"], b0, self.cout), self.cout, stream), "cp_wino_output_transform_f32(%s)" % self.name)

    def micro_steps(self):
        """[(tag, fn(stream))] of one run(): "H" = HBM-bound transform passes, "M" = the matrix-pipe GEMM (the two-stream forward puts them on
        different streams).  Only without batch chunking (one chunk)."""
        lib = _lib.load()
        e = self.epi
        tp = self.tiles(self.batch, self.h, self.w, self.dil)[1]
        out: List = []

        def t_in(stream):
            off = 0
            for (cpad, _), s in zip(self.sources, self.srcs):
                self._input_transform(s["data"].data_ptr(), s["ld"], cpad, self.batch, off, stream)
                off += cpad

        def t_gemm(stream):
            self.run_gemm(stream, tp)

        def t_out(stream):
            nxt = getattr(self, "fuse_next", None)
            if nxt is not None:
                sc_, sh_ = self._tables_for_next()
                with self._armed(nxt.mon_ptr if nxt.planes == _lib.PLANES_F16X2 else None):
                    check(lib.cp_wino_output_input_transform_f32(self.M.data_ptr(), self.cout, self.batch, self.h, self.w, self.dil, _ptr(e["residual"]), self.cout,
                                                                 sc_, sh_, e["act"], _ptr(e["out_raw"]), self.cout, None, self.cout,
                                                                 self.V.data_ptr(), nxt.ktot, 0, stream), "cp_wino_output_input_transform_f32(%s)" % self.name)
            else:
                check(lib.cp_wino_output_transform_f32(self.M.data_ptr(), self.cout, self.batch, self.h, self.w, self.dil, _ptr(e["residual"]), self.cout,
                                                       _ptr(e["scale"]), _ptr(e["shift"]), _ptr(e["epi_label"]), e["act"], _ptr(e["out_raw"]), self.cout,
                                                       _ptr(e["out_act"]), self.cout, stream), "cp_wino_output_transform_f32(%s)" % self.name)

        if not getattr(self, "skip_input", False):
            out.append(("H", t_in))
        out.append(("M", t_gemm))
        out.append(("H", t_out))
        return out

    def run_gemm(self, stream: int, tp: Optional[int] = None):
        lib = _lib.load()
        if tp is None:  # stand-alone timing of the GEMM (bench.py): every chunk's GEMM back to back on the same scratch
            for _, _, tpc in self.chunks():
                self.run_gemm(stream, tpc)
            return
        if WINO_GROUPED_CONV:  # the grouped mode of the general conv kernel (kept for comparison; whole batch only)
            check(lib.cp_conv2d_fwd_f32(C.byref(self.desc), stream), "cp_conv2d_fwd_f32(wino %s)" % self.name)
        else:
            if self.Us is not None:
                check(lib.cp_wino_gemm_split_scaled_f32(self.V.data_ptr(), self.Us.data_ptr(), self.M.data_ptr(), 36 * tp, tp, self.ktot, self.cout, self.planes,
                                                        self.c_scale / self.v_scale, stream), "cp_wino_gemm_split_scaled_f32(%s)" % self.name)
            else:
                check(lib.cp_wino_gemm_f32(self.V.data_ptr(), self.U.data_ptr(), self.M.data_ptr(), 36 * tp, tp, self.ktot, self.cout, stream),
                      "cp_wino_gemm_f32(%s)" % self.name)

    @property
    def flops(self) -> float:
        """FLOPs of the convolution this launch group REPLACES (2*M*9*Cin*Cout), the figure the per-layer tables quote."""
        cin = sum(s[1] for s in self.sources)
        return 2.0 * self.batch * self.h * self.w * 9 * cin * self.cout

    @property
    def gemm_flops(self) -> float:
        """FLOPs the grouped GEMMs actually execute (36 planes x padded tiles, summed over the batch groups)."""
        return sum(2.0 * 36 * tp * self.ktot * self.cout for _, _, tp in self.chunks())


WINO_DIRECT_UNDILATED_128 = os.environ.get("CASAPOSE_WINO_DIRECT_128", "")   # "" = by conv mode (f16x2: yes), "0" / "1" force (A/B)
WINO_MIN_K = int(os.environ.get("CASAPOSE_WINO_MIN_K", "0"))      # 0 = the measured defaults below
WINO_MIN_COUT = int(os.environ.get("CASAPOSE_WINO_MIN_COUT", "128"))


def wino_eligible(kh: int, stride: int, dilation: int, pad: int, sources: Sequence[Tuple[int, int]], cout: int, split_gemm: bool = False,
                  f16x2: bool = False) -> bool:
    """Measured on MI355X (profiles/): the 4x MFMA saving beats the two transform passes once the GEMM is deep and wide enough -- K >= 256 with
    the fp32-MFMA GEMM; K >= 128 when the GEMM runs on the bf16 pipe (split_gemm: the dilated 128 -> 256 layer stage3_unit1_conv1, which no
    split kernel covers directly, 0.43 -> 0.17 ms at bs 16; the three 128 -> 128 stage-2 layers 0.37 -> 0.36 ms).  f16x2 (three products per fp32
    product): the UNDILATED 128-channel layers are faster on the direct split kernel (stage 2: 0.27 against 0.30 ms, 1882 against 1872 images/s) --
    their Winograd GEMM is HBM-bound on V and M -- so K >= 256 again for them; the dilated 128 -> 256 layer has no direct kernel and stays."""
    k = sum(s[0] for s in sources)
    min_k = WINO_MIN_K or (128 if split_gemm else 256)
    direct_128 = f16x2 if WINO_DIRECT_UNDILATED_128 == "" else WINO_DIRECT_UNDILATED_128 == "1"
    if direct_128 and dilation == 1 and split_gemm and not WINO_MIN_K:
        min_k = 256
        # Round 6 (tools/debug/wino_parts.py, bs 16): decoder block 2 (384 -> 128, undilated) takes 88 + 87 + 25 us as input transform + GEMM + output
        # transform and 178 us on the direct f16x2 kernel (the same shape with a tap mask, block 7, runs there already): with <= 128 output channels
        # the GEMM's M traffic is small but V (2.25 x the 384-channel input in fp32) is not.  Undilated layers of that width go direct whatever K.
        if cout <= 128:
            return False
    return kh == 3 and stride == 1 and pad == dilation and k >= min_k and k % 32 == 0 and cout >= WINO_MIN_COUT and cout % 4 == 0


class ForwardPlan:
    """All buffers, descriptors and the launch order for one (batch, H, W) input shape."""

    def __init__(self, net: "CasaposeNet", batch: int, h: int, w: int, fuse_upsample: bool = True, fuse_heads: bool = True):
        if h % 8 or w % 8:
            raise ValueError("input height/width must be multiples of 8 (got %dx%d)" % (h, w))
        self.net, self.batch, self.h, self.w = net, batch, h, w
        self.fuse_upsample = fuse_upsample
        self.fuse_heads = fuse_heads and net.decoder_dims[4] == 32 and not net.pvnet
        dev = net.device
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        B = batch
        K, V = net.seg_dim, net.ver_dim
        self.out_ld = K + V
        self.steps: List = []  # callables taking (stream)
        self.convs: List[FusedConv] = []
        # f16x2 range guard: the first run() of this plan goes layer by layer, measures what every f16x2 layer is about to convert and demotes the
        # layers outside the range condition to the exact split (ForwardPlan._calibrate); f16x2_report keeps what it saw
        self.needs_calibration = bool(net.conv_planes == _lib.PLANES_F16X2 and net.f16x2_guard)
        self.f16x2_report: Dict[str, Tuple[float, str]] = {}
        # ... and afterwards every forward runs armed; every F16X2_MONITOR_EVERY-th the slots are read back and judged without a synchronisation (_poll_monitor)
        self._mon = self._mon_host = self._mon_event = None
        self._since_monitor = 0
        self.monitor_checks = self.monitor_fired = 0   # armed forwards judged so far / how many of them re-armed the calibration
        lib = _lib.load()
        P = net.device_tables
        hs = [h, h // 2, h // 4, h // 8]
        ws = [w, w // 2, w // 4, w // 8]

        def new(*shape):
            return torch.empty(*shape, **f32)

        # ---- buffers that are (re)bound per call ------------------------------------------
        self.img4 = new(B, h, w, 4)
        self.labels = [torch.empty(B, hs[l], ws[l], **u8) for l in range(4)]
        self.labels_from_head = False   # True: block 5's fused segmentation head writes labels[0] in its epilogue
        self.pnorm = [new(B, hs[l], ws[l]) for l in range(4)]
        self.sel = [torch.empty(B, hs[l], ws[l], **u8) for l in range(3)]
        self.sel_zero = [torch.zeros(B, hs[l], ws[l], **u8) for l in range(3)]  # plain nearest x2 = "guided" with neighbour 0 everywhere
        self.gmask = [torch.empty(B, hs[l], ws[l], **u8) if any(net.bilinear) else None for l in range(3)]  # GuidedBilinearUpsampling match masks
        self._out_bound: List[Tuple[FusedConv, int, str]] = []  # (conv, channel offset, descriptor field) writing into the per-call output
        self._bufs: List[torch.Tensor] = []

        self.wino_V = self.wino_M = None

        def conv(layer: FusedConv, act_private: bool = False, **kw):
            """act_private: nothing but the NEXT convolution of the plan reads this layer's activated output (lets two consecutive Winograd
            layers hand it over on chip: WinoConv.fuse_next)"""
            wl = net.wino_by_name.get(layer.name) if net.use_winograd else None
            plain = all(s.get("mode", _lib.SRC_DIRECT) == _lib.SRC_DIRECT and not s.get("pre") for s in kw["srcs"])
            # bf16 conv mode: the deep 3x3 layers (cout a multiple of 128, any of the dilations, no labels) on the direct bf16-operand kernel
            # (csrc/conv_bf16d.hip) instead of the two-plane Winograd path: measured 1.15-1.6x per layer at bs 16 (CASAPOSE_BF16_DEEP=0: Winograd)
            if (net.conv_planes == 1 and BF16_DEEP and layer.wp_split_f32 is not None and plain and layer.kh == 3 and layer.cout % 128 == 0
                    and kw.get("tap_label") is None and kw.get("epi_label") is None and kw.get("head_out") is None and kw.get("stride", 1) == 1):
                layer.bind(batch=B, **kw)
                if lib.cp_conv_bf16_deep_applicable(C.byref(layer.desc)):
                    layer.deep_bf16 = True
                    self.convs.append(layer)
                    self.steps.append(layer.run)
                    return
            if (wl is not None and plain and kw.get("tap_label") is None and kw.get("head_out") is None and kw.get("stride", 1) == 1
                    and kw.get("out_raw_ld") is None and kw.get("out_act_ld") is None and all(s["ld"] == c[0] for s, c in zip(kw["srcs"], wl.sources))):
                _, tp = WinoConv.tiles(B, kw["in_h"], kw["in_w"], kw.get("dilation", 1))
                nv, nm = 36 * tp * wl.ktot, 36 * tp * wl.cout
                if self.wino_V is None or self.wino_V.numel() < nv:
                    self.wino_V = torch.empty(nv, **f32)
                if self.wino_M is None or self.wino_M.numel() < nm:
                    self.wino_M = torch.empty(nm, **f32)
                self._wino_pending.append((wl, dict(batch=B, in_h=kw["in_h"], in_w=kw["in_w"], dilation=kw.get("dilation", 1), srcs=kw["srcs"],
                                                    residual=kw.get("residual"), scale=kw.get("scale"), shift=kw.get("shift"), epi_label=kw.get("epi_label"),
                                                    act=kw.get("act", 0), out_raw=kw.get("out_raw"), out_act=kw.get("out_act")), bool(act_private)))
                self.convs.append(wl)
                self.steps.append(wl.run)
                return
            layer.bind(batch=B, **kw)
            # conv mode (CASAPOSE_INFER_CONV_MODE / CasaposeNet(conv_mode=...), default "split"): the 3x3 / stride-1 layers off the Winograd path on the
            # bf16 matrix pipe (csrc/conv_hsplit.hip), 3 planes = exact three-way split (fp32-equivalent), 1 plane = bf16 operands; layers outside
            # its range (stem, strided and dilated layers, ...) keep the fp32-MFMA kernels
            layer.stem_split = 0
            planes = net.planes_for(layer.name)   # the net's conv mode, or the exact split for a layer the f16x2 guard has demoted
            if planes and layer.wp_split_f32 is not None and lib.cp_conv_split_applicable(C.byref(layer.desc)):
                layer.split_mode = planes
            elif (planes and STEM_SPLIT and layer.wp_stem_split_f32 is not None and kw.get("stride", 1) == 2 and kw.get("pad", 0) == 3
                  and kw.get("dilation", 1) == 1 and lib.cp_conv_selected_tile(C.byref(layer.desc)) == _lib.TILE_STEM):
                layer.stem_split = planes   # conv0 on the bf16 matrix pipe (exact split / bf16 operands), csrc/conv_stem_split.hip
            elif planes and layer.kh == 1 and layer.name + ".kernel" in net.params:
                layer.enable_gemm_split(np.asarray(net.params[layer.name + ".kernel"], np.float32), planes if planes in (3, _lib.PLANES_F16X2) else 2)
            self.convs.append(layer)
            self.steps.append(layer.run)

        self._wino_pending: List = []  # bound at the end, when the shared scratch has its final size

        L = net.layers_by_name
        # ---- encoder (resnet.py:246-305) ---------------------------------------------------
        x2s = new(B, hs[1], ws[1], 64)
        conv(L["conv0"], in_h=h, in_w=w, stride=2, pad=3,
             srcs=[dict(data=self.img4, ld=4, pre=P["bn_data"])],
             scale=P["bn0"][0], shift=P["bn0"][1], act=_lib.ACT_RELU, out_act=x2s)
        a = new(B, hs[2], ws[2], 64)
        s1, b1 = P["stage1_unit1_bn1"]

        def pool(stream, src=x2s, dst=a, s1=s1, b1=b1):
            check(lib.cp_maxpool3x3s2_f32(src.data_ptr(), B, hs[1], ws[1], 64, s1.data_ptr(), b1.data_ptr(), 1,
                                          dst.data_ptr(), stream), "cp_maxpool3x3s2_f32")

        self.steps.append(pool)
        self._bufs += [x2s, a]
        cur_h, cur_w, cin = hs[2], ws[2], 64
        x_raw = None
        taps: Dict[str, torch.Tensor] = {"x2s": x2s}
        tap_names = ["x4s", "x8s", "x16s", "x32s"]
        for s, f in enumerate(STAGE_FILTERS):
            d = STAGE_DILATION[s]
            for u in range(2):
                base = "stage%d_unit%d_" % (s + 1, u + 1)
                stride = STAGE_STRIDE[s] if u == 0 else 1
                oh, ow = (cur_h - 1) // stride + 1, (cur_w - 1) // stride + 1
                t = new(B, oh, ow, f)
                bn2 = P[base + "bn2"]
                if u == 0:
                    sc = new(B, oh, ow, f)
                    conv(L[base + "sc"], in_h=cur_h, in_w=cur_w, stride=stride, srcs=[dict(data=a, ld=cin)], out_raw=sc)
                    shortcut = sc
                else:
                    shortcut = x_raw
                conv(L[base + "conv1"], in_h=cur_h, in_w=cur_w, stride=stride, dilation=d, pad=d, act_private=True,   # t feeds conv2 only
                     srcs=[dict(data=a, ld=cin)], scale=bn2[0], shift=bn2[1], act=_lib.ACT_RELU, out_act=t)
                if u == 0:
                    nxt = P["stage%d_unit2_bn1" % (s + 1)]
                    x_raw = new(B, oh, ow, f)
                else:
                    nxt = P["stage%d_unit1_bn1" % (s + 2)] if s < 3 else P["bn1"]
                    x_raw = None
                a_next = new(B, oh, ow, f)
                # unit 1's activated output feeds unit 2's conv1 only (its shortcut is the RAW sum); unit 2's also feeds the next stage's 1x1
                # shortcut / the decoders / the backbone's taps
                conv(L[base + "conv2"], in_h=oh, in_w=ow, dilation=d, pad=d, srcs=[dict(data=t, ld=f)], act_private=(u == 0),
                     residual=shortcut, out_raw=x_raw, scale=nxt[0], shift=nxt[1], act=_lib.ACT_RELU, out_act=a_next)
                self._bufs += [t, shortcut, a_next]
                a, cur_h, cur_w, cin = a_next, oh, ow, f
                if u == 1:
                    taps[tap_names[s]] = a
        self.taps = taps
        self.encoder_steps = len(self.steps)   # run_encoder() stops here: the bare ResNet-18 backbone (resnet.py:319: the model's five outputs)
        x32s, x8s, x4s = taps["x32s"], taps["x8s"], taps["x4s"]
        skips = [None, (x8s, 128), (x4s, 64), (x2s, 64), (self.img4, 4)]
        dims = net.decoder_dims
        lvl = [3, 3, 2, 1, 0]  # pyramid level each decoder block runs at

        # ---- decoder 1 (pose_models.py:541-546) -------------------------------------------
        self.out = None  # bound per call
        prev, prev_c = None, 0
        for i in range(5):
            name = "pv_block_%d_conv2d" % (i + 1)
            bn = P["pv_block_%d_bn" % (i + 1)]
            l = lvl[i]
            o = new(B, hs[l], ws[l], dims[i])
            if i == 0:
                srcs = [dict(data=x32s, ld=512)]
            else:
                up = i >= 2  # blocks 2,3,4 upsample their OUTPUT (casapose.py:109-140): consumed by block i+1
                src0 = prev
                mode = _lib.SRC_DIRECT
                if up:
                    # round 3 materialised the x2 bilinear tensor for the 32-output-channel layers (blocks 4, 5) on the bf16 pipe: four taps per
                    # halo pixel from global memory did not fit their loaders (0.77 / 1.29 ms fused against 0.46 / 0.63 ms direct).  The loaders
                    # now stage the half-resolution tile in LDS and interpolate from there (csrc/conv_hsplit.hip), so the fused form is the
                    # default everywhere; CASAPOSE_MATERIALISE_BILINEAR=1 restores the separate streaming pass (A/B measurements)
                    if fuse_upsample and not (net.conv_planes and dims[i] <= 32 and MATERIALISE_BILINEAR):
                        mode = _lib.SRC_BILINEAR_X2
                    else:
                        big = new(B, hs[l], ws[l], prev_c)
                        self.steps.append(self._bilinear_step(prev, big, hs[l] // 2, ws[l] // 2, prev_c))
                        self._bufs.append(big)
                        src0 = big
                srcs = [dict(data=src0, ld=prev_c, mode=mode), dict(data=skips[i][0], ld=skips[i][1])]
            fused = self.fuse_heads and i == 4
            if fused:  # blocks 5 + pv_final_conv_segmentation in one launch; the 32-channel tensor is never stored
                L[name].attach_head(net.params["pv_final_conv_segmentation.kernel"])
                # ... and the hard label map (arg-max of the K logits) straight from the head's registers: no pass over the strided records
                conv(L[name], in_h=hs[l], in_w=ws[l], pad=1, srcs=srcs, scale=bn[0], shift=bn[1], act=_lib.ACT_LEAKY01,
                     head_out=self.img4, head_out_ld=self.out_ld, head_label_out=self.labels[0], head_label_classes=K)
                self._out_bound.append((L[name], 0, "head_out"))
                self.labels_from_head = True
            else:
                extra = {}
                if i == 0 and net.reuse_first:
                    self.y_raw = new(B, hs[l], ws[l], dims[0])
                    extra = dict(out_raw=self.y_raw)
                conv(L[name], in_h=hs[l], in_w=ws[l], pad=1, srcs=srcs, scale=bn[0], shift=bn[1],
                     act=_lib.ACT_RELU if i == 0 else _lib.ACT_LEAKY01, out_act=o, **extra)
            self._bufs.append(o)
            prev, prev_c = o, dims[i]
        self.fuse_head2 = False
        if net.pvnet:  # one head for all K + ver_dim channels; no conditioning, no second decoder
            head = L["pv_final_conv"]
            conv(head, in_h=h, in_w=w, srcs=[dict(data=prev, ld=prev_c)], out_raw=self.img4, out_raw_ld=self.out_ld)
            self._out_bound.append((head, 0, "out_raw"))
            self.seg_input_ptr = None
            self._bind_winograd()
            return
        if not self.fuse_heads:
            seg_head = L["pv_final_conv_segmentation"]
            conv(seg_head, in_h=h, in_w=w, srcs=[dict(data=prev, ld=prev_c)], out_raw=self.img4, out_raw_ld=self.out_ld)
            self._out_bound.append((seg_head, 0, "out_raw"))

        # ---- hard label map + pyramid (pose_models.py:547-559) -----------------------------
        self.seg_input_ptr = None  # set per call when the model has a data_segmentation input

        def label_step(stream):
            if self.seg_input_ptr is not None:
                src, ld = self.seg_input_ptr, K
            else:
                src, ld = self.out.data_ptr(), self.out_ld
            if self.seg_input_ptr is not None or not self.labels_from_head:   # else block 5's fused head has written labels[0] already
                check(lib.cp_argmax_labels(src, ld, K, B * h * w, self.labels[0].data_ptr(), stream), "cp_argmax_labels")
            lab = (C.c_void_p * 4)(*[t.data_ptr() for t in self.labels])
            pn = (C.c_void_p * 4)(*[t.data_ptr() for t in self.pnorm])
            sl = (C.c_void_p * 3)(*[t.data_ptr() for t in self.sel])
            check(lib.cp_label_pyramid(self.labels[0].data_ptr(), B, h, w, lab, pn, sl, stream), "cp_label_pyramid")
            for l_ in range(3):
                if self.gmask[l_] is not None:
                    check(lib.cp_guided_match_mask(self.labels[l_].data_ptr(), self.labels[l_ + 1].data_ptr(), B, hs[l_], ws[l_], self.gmask[l_].data_ptr(), stream),
                          "cp_guided_match_mask")

        self.steps.append(label_step)

        # ---- decoder 2 (pose_models.py:561-616) -------------------------------------------
        prev, prev_c = None, 0
        for i in range(5):
            partial = net.partial[i]
            name = ("pv_block_%d_prepare_conv2d" if partial else "pv_block_%d_conv2d") % (i + 6)
            tab = P["pv_block_%d_clade" % (i + 6)]
            l = lvl[i]
            o = new(B, hs[l], ws[l], dims[i])
            if i == 0 and net.reuse_first:  # casa_layer(y, "6", skip_conv=True): CLADE + ReLU on block 1's raw convolution output
                def clade_step(stream, src=self.y_raw, dst=o, tab=tab, lab=self.labels[l], n=B * hs[l] * ws[l], c=dims[0]):
                    check(lib.cp_affine_act_f32(src.data_ptr(), n, c, c, tab[0].data_ptr(), tab[1].data_ptr(), lab.data_ptr(), _lib.ACT_RELU,
                                                dst.data_ptr(), c, stream), "cp_affine_act_f32(pv_block_6_clade)")

                self.steps.append(clade_step)
                self._bufs.append(o)
                prev, prev_c = o, dims[i]
                continue
            if i == 0:
                srcs = [dict(data=x32s, ld=512)]
            else:
                up = i >= 2
                src0, mode, sel = prev, _lib.SRC_DIRECT, None
                if up and net.bilinear[i - 1]:
                    if not net.guided[i - 1]:
                        raise NotImplementedError("bilinear_upsampling without guided_upsampling in decoder 2 is not built")
                    big = new(B, hs[l], ws[l], prev_c)  # GuidedBilinearUpsampling: a 4-tap blend, materialised
                    self.steps.append(self._guided_bilinear_step(prev, self.gmask[l], big, hs[l] // 2, ws[l] // 2, prev_c))
                    self._bufs.append(big)
                    src0 = big
                elif up:
                    selmap = self.sel[l] if net.guided[i - 1] else self.sel_zero[l]  # block i-1 upsampled its output (casapose.py:109-131)
                    if fuse_upsample:
                        mode, sel = _lib.SRC_NEAREST_SEL, selmap
                    else:
                        big = new(B, hs[l], ws[l], prev_c)
                        self.steps.append(self._guided_step(prev, selmap, big, hs[l] // 2, ws[l] // 2, prev_c))
                        self._bufs.append(big)
                        src0 = big
                srcs = [dict(data=src0, ld=prev_c, mode=mode, sel=sel), dict(data=skips[i][0], ld=skips[i][1])]
                if not net.skips2:
                    srcs = srcs[:1]
            pk = dict(tap_label=self.labels[l], row_scale=self.pnorm[l]) if partial else {}
            # the fused head lives in the halo kernel, which gathers a guided/nearest x2 source only together with the tap mask
            fused = self.fuse_heads and i == 4 and (partial or not fuse_upsample or net.bilinear[3])
            self.fuse_head2 = fused if i == 4 else False
            if fused:  # block 10 + pv_final_conv_vertex in one launch
                L[name].attach_head(net.params["pv_final_conv_vertex.kernel"])
                conv(L[name], in_h=hs[l], in_w=ws[l], pad=1, srcs=srcs, scale=tab[0], shift=tab[1], epi_label=self.labels[l],
                     act=_lib.ACT_LEAKY01, head_out=self.img4, head_out_ld=self.out_ld, **pk)
                self._out_bound.append((L[name], K, "head_out"))
            else:
                conv(L[name], in_h=hs[l], in_w=ws[l], pad=1, srcs=srcs, scale=tab[0], shift=tab[1], epi_label=self.labels[l],
                     act=_lib.ACT_RELU if i == 0 else _lib.ACT_LEAKY01, out_act=o, **pk)
            self._bufs.append(o)
            prev, prev_c = o, dims[i]
        if not self.fuse_head2:
            ver_head = L["pv_final_conv_vertex"]
            conv(ver_head, in_h=h, in_w=w, srcs=[dict(data=prev, ld=prev_c)], out_raw=self.img4, out_raw_ld=self.out_ld)
            self._out_bound.append((ver_head, K, "out_raw"))
        self._whole_records(L, K, B * h * w, new)
        self._bind_winograd()

    def _bind_winograd(self):
        lib = _lib.load()
        for wl, kw, _ in self._wino_pending:
            wl.bind(V=self.wino_V, M=self.wino_M, **kw)
            wl.fuse_next, wl.skip_input = None, False
        # consecutive Winograd layers A -> B of one geometry where B's only source is A's activated output and nothing else reads it: A's output
        # transform writes B's transformed input directly (cp_wino_output_input_transform_f32); the activated map is not stored
        for (a, ka, private), (b, kb, _) in zip(self._wino_pending, self._wino_pending[1:]):
            if not (WINO_FUSE_OUT_IN and private and WINO_CHUNK == 0 and not WINO_GROUPED_CONV):
                continue
            ia, ib = self.steps.index(a.run), self.steps.index(b.run)
            same = all(ka[k] == kb[k] for k in ("batch", "in_h", "in_w", "dilation"))
            if (ib == ia + 1 and same and len(kb["srcs"]) == 1 and kb["srcs"][0]["data"] is ka["out_act"] and ka["out_act"] is not None
                    and ka["epi_label"] is None and b.ktot == a.cout
                    and lib.cp_wino_output_input_applicable(ka["batch"], ka["in_h"], ka["in_w"], ka["dilation"], a.cout)):
                a.fuse_next, b.skip_input = b, True
        self._wino_pending = []

    # ---- f16x2 range guard (the C-ABI side: cp_f16x2_monitor_set / cp_f16x2_range_check / cp_amax_f32) -----------------------------
    def _run_armed(self, stream: int, fresh: bool = True):
        """One forward with every f16x2 layer reporting into its monitor slot (slot i <-> self.convs[i]; include/casapose_hip.h): the converting
        kernels fold max |x| of what they convert -- a direct layer's staged sources, the stem's image through its input affine, a 1x1 GEMM's rows,
        the planes V a Winograd transform writes, the activated map a fused head converts in registers -- into the slot with one atomic per wave.
        fresh = False keeps what earlier armed forwards left in the slots (the monitor's sticky maxima)."""
        if self._mon is None:
            self._mon = torch.zeros(4 * len(self.convs), dtype=torch.int32, device=self.net.device)
            self._mon_host = torch.zeros(4 * len(self.convs), dtype=torch.int32).pin_memory()
        elif fresh:
            self._mon.zero_()
        base = self._mon.data_ptr()
        only = os.environ.get("CASAPOSE_F16X2_MONITOR_ONLY", "")   # (measurement aid: arm one kernel family only -- "wino", "fused" or a layer-name prefix)
        for i, c in enumerate(self.convs):
            c.mon_ptr = base + 16 * i
            if only and not ((only == "wino" and isinstance(c, WinoConv)) or (only == "fused" and isinstance(c, FusedConv)) or c.name.startswith(only)):
                c.mon_ptr = None
        try:
            for step in self.steps:
                step(stream)
        finally:
            for c in self.convs:   # (never left attached: a layer run outside a plan's forward must not write into a plan's slots)
                c.mon_ptr = None

    def _judge(self, words: np.ndarray, lo: float, hi: float):
        """[(layer, "in" | "head", max |x| as converted, status, power of two)] for every slot that reported; status as cp_f16x2_range_check:
        0 inside [lo, hi], 1 rescale by the power of two, 2 no power of two helps (or not finite)"""
        lib = _lib.load()
        w = np.ascontiguousarray(words).view(np.uint32).reshape(-1, 4)
        out = []
        for i, c in enumerate(self.convs):
            if int(w[i, 1]) == 0:
                continue
            for kind, bits in (("in", w[i, 0]), ("head", w[i, 2])):
                if kind == "head" and (bits == 0 or isinstance(c, WinoConv)):
                    continue
                amax = float(np.array([bits], np.uint32).view(np.float32)[0])
                r = C.c_float(1.0)
                st = lib.cp_f16x2_range_check(amax, lo, hi, C.byref(r))
                out.append((c, kind, amax, st, float(r.value)))
        return out

    def _producer_of(self, wl: "WinoConv") -> Optional["WinoConv"]:
        for c in self.convs:
            if isinstance(c, WinoConv) and c.fuse_next is wl:
                return c
        return None

    def _calibrate(self, stream: int):
        """Fit every f16x2 layer to what it converts: armed forwards (_run_armed), one host read per pass, exact remedies, repeated until a pass
        finds every layer inside [F16X2_AMAX_LO, F16X2_AMAX_HI] (a layer downstream of a clamping one is measured again once that one is fixed;
        the bench network settles in one pass, adversarial statistics in two or three).  Three places convert fp32 activations to fp16 pairs:
          * a Winograd layer's GEMM converts V = B^T d B (measured where it is written: the layer's own input transform or the producing layer's
            fused output -> input transform): the transform multiplies by a power of two (WinoConv.v_scale: exact) and the GEMM's accumulator
            factor undoes it -- the weights' remedy (cp_f16x2_weight_scale) applied to the activations;
          * a fused 1x1 head converts the activated 32-channel map that never reaches HBM (measured in the epilogue's registers): the normalisation
            table feeding the activation is multiplied by the power of two (ReLU / leaky ReLU are positively homogeneous: act(s x) = s act(x)
            exactly) and the head's accumulator factor undoes it (FusedConv.head_in_scale);
          * the direct kernels, the stem and the 1x1 GEMMs convert stored tensors that other layers read too: the layer runs on the exact
            three-way bf16 split (no range condition) -- for this plan and, through net.f16x2_fallback, for later plans (CasaposeNet.recalibrate()
            forgets that).
        The last pass IS the forward whose output run() returns.  Afterwards the monitor keeps watching (run(), _poll_monitor)."""
        net = self.net
        lo, hi = F16X2_AMAX_LO, F16X2_AMAX_HI
        changed: Dict[str, str] = {}

        def note(name: str, amax: float, what: str, action: str):
            self.f16x2_report[name] = (amax, action)
            if action != "f16x2":
                changed[name] = "%s (max |%s| = %.3g: %s)" % (name, what, amax, action)

        settled = False
        for _ in range(8):
            self._run_armed(stream)
            words = self._mon.cpu().numpy()   # (synchronises: calibration only)
            acted = False
            for c, kind, amax, st, r in self._judge(words, lo, hi):
                if isinstance(c, WinoConv):
                    a0 = amax / c.v_scale   # as the un-scaled transform would write it
                    prod = self._producer_of(c) if c.skip_input else None
                    can_scale = (not c.skip_input) or (prod is not None and prod.epi["scale"] is not None)
                    if st == 1 and can_scale:
                        c.v_scale *= r
                        acted = True
                    elif st != 0:
                        c.demote_to_exact_split()
                        c.v_scale = 1.0
                        acted = True
                    note(c.name, a0, "V", "exact bf16 split" if c.planes != _lib.PLANES_F16X2 else ("f16x2" if c.v_scale == 1.0 else "f16x2, V x %g" % c.v_scale))
                elif kind == "in":
                    if st != 0:
                        net.f16x2_fallback[c.name] = "max |a| = %.3g outside [%g, %g]" % (amax, lo, hi)
                        c.demote_to_exact_split(net.params.get(c.name + ".kernel"))
                        note(c.name, amax, "a", "exact bf16 split")
                        self.f16x2_report.pop(c.name + ":head", None)
                        acted = True
                    else:
                        note(c.name, amax, "a", "f16x2")
                elif c.name not in net.f16x2_fallback:   # the fused head's operand (a layer demoted in this very pass no longer converts it)
                    a0 = amax / c.head_in_scale
                    if st == 1:
                        c.set_head_in_scale(c.head_in_scale * r)
                        acted = True
                    elif st != 0:
                        net.f16x2_fallback[c.name] = "head operand max = %.3g" % a0
                        c.demote_to_exact_split(net.params.get(c.name + ".kernel"))
                        note(c.name, a0, "head a", "exact bf16 split")
                        acted = True
                        continue
                    note(c.name + ":head", a0, "head a", "f16x2" if c.head_in_scale == 1.0 else "f16x2, head input x %g" % c.head_in_scale)
            if not acted:
                settled = True
                break
        self.needs_calibration = False
        self._since_monitor, self._mon_event = 0, None
        self._mon.zero_()   # the monitor's window starts behind the calibration (a sticky maximum of the calibration batch would hide a later, smaller regime)
        import warnings
        if not settled:
            warnings.warn("conv_mode f16x2: the range calibration did not settle in 8 passes; the plan keeps its last fit")
        if changed and not net._f16x2_warned:
            net._f16x2_warned = True
            warnings.warn("conv_mode f16x2: %d layer(s) outside the fp16 range condition [%g, %g] were rescaled by a power of two or moved to the exact bf16 split: %s"
                          % (len(changed), lo, hi, "; ".join(changed.values())))

    def _poll_monitor(self):
        """judge an armed forward whose slots have arrived in pinned host memory (never waits): a layer outside [LO / SLACK, HI * SLACK] -- a batch far
        from the one the plan was calibrated on -- re-arms the calibration, with a warning"""
        ev = self._mon_event
        if ev is None or not ev.query():
            return
        self._mon_event = None
        lo, hi = F16X2_AMAX_LO / F16X2_MONITOR_SLACK, F16X2_AMAX_HI * F16X2_MONITOR_SLACK
        out = [(c.name + (":head" if kind == "head" else ""), amax) for c, kind, amax, st, _ in self._judge(self._mon_host.numpy().copy(), lo, hi) if st != 0]
        self.monitor_checks += 1
        if out:
            self.monitor_fired += 1
            self.needs_calibration = True
            self.net._f16x2_warned = False
            import warnings
            warnings.warn("conv_mode f16x2: the range monitor found %d layer(s) converting operands outside [%g, %g] (%s): this batch is far from the one the "
                          "plan was calibrated on; calibrating again" % (len(out), lo, hi, "; ".join("%s max %.3g" % o for o in out[:6])))

    def recalibrate(self):
        """calibrate again on the next forward (keeps what net.f16x2_fallback has demoted: CasaposeNet.recalibrate() forgets that too)"""
        self.needs_calibration = bool(self.net.conv_planes == _lib.PLANES_F16X2 and self.net.f16x2_guard)

    def _device_amax(self, t: torch.Tensor) -> float:
        """max |t| through cp_amax_f32 (diagnostics; synchronises)"""
        slot = torch.zeros(4, dtype=torch.int32, device=t.device)
        check(_lib.load().cp_amax_f32(t.data_ptr(), 1, 0, t.numel(), slot.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream), "cp_amax_f32")
        return float(slot[:1].cpu().numpy().view(np.float32)[0])

    def f16x2_operand_ranges(self) -> Dict[str, Optional[Tuple[float, float]]]:
        """The range condition of the fp16 two-way split made checkable.  Call after a forward: for every layer this plan runs in
        f16x2, (max |a| over the layer's source tensors, upper bound of the magnitude its split converts) -- the bound is max |a| itself for the
        direct kernels (an interpolated / selected source is a convex combination of the stored one; the stem's input affine is applied), 100 x
        max |a| for a Winograd layer (V = B^T d B: the rows of B^T sum to at most 10 in magnitude).  fp32-level accuracy needs the first number
        >= ~1 and the second <= 65504.  None: the layer's input exists only inside a fused output -> input transform (it is the activated,
        normalised output of the layer before it)."""
        out: Dict[str, Optional[Tuple[float, float]]] = {}
        for conv in self.convs:
            if hasattr(conv, "gemm_flops"):   # WinoConv
                if conv.planes != _lib.PLANES_F16X2:
                    continue
                if conv.skip_input:
                    out[conv.name] = None
                    continue
                amax = max(self._device_amax(s["data"]) for s in conv.srcs)
                out[conv.name] = (amax, 100.0 * amax)
                continue
            planes = conv.split_mode or conv.stem_split or ((conv._gemm or {}).get("planes", 0))
            if planes != _lib.PLANES_F16X2:
                continue
            amax = 0.0
            for sdict in conv._srcs:
                a = self._device_amax(sdict["data"])
                pre = sdict.get("pre")
                if pre:
                    a = a * self._device_amax(pre[0]) + self._device_amax(pre[1])
                amax = max(amax, a)
            out[conv.name] = (amax, amax)
        return out

    def _bilinear_step(self, src, dst, sh, sw, c):
        lib = _lib.load()
        B = self.batch

        def step(stream):
            check(lib.cp_upsample_bilinear_x2_f32(src.data_ptr(), B, sh, sw, c, dst.data_ptr(), stream), "cp_upsample_bilinear_x2_f32")

        return step

    def _guided_bilinear_step(self, src, mask, dst, sh, sw, c):
        lib = _lib.load()
        B = self.batch

        def step(stream):
            check(lib.cp_guided_bilinear_upsample_x2_f32(src.data_ptr(), mask.data_ptr(), B, sh, sw, c, dst.data_ptr(), stream), "cp_guided_bilinear_upsample_x2_f32")

        return step

    def _guided_step(self, src, sel, dst, sh, sw, c):
        lib = _lib.load()
        B = self.batch

        def step(stream):
            check(lib.cp_guided_upsample_x2_f32(src.data_ptr(), sel.data_ptr(), B, sh, sw, c, dst.data_ptr(), stream), "cp_guided_upsample_x2_f32")

        return step

    def conv_flops(self) -> float:
        return sum(c.flops for c in self.convs)

    def run_encoder(self, img: torch.Tensor) -> List[torch.Tensor]:
        """The encoder alone: the five taps of the reference's backbone model in its output order (resnet.py:251-252,291,303-305,319) --
        relu0 [B,H/2,W/2,64], stage2_unit1_relu1 [B,H/4,W/4,64], stage3_unit1_relu1 [B,H/8,W/8,128], stage4_unit1_relu1 [B,H/8,W/8,256],
        relu1 [B,H/8,W/8,512].  The tensors are the plan's own buffers (valid until the next run)."""
        B, h, w = self.batch, self.h, self.w
        if tuple(img.shape) != (B, h, w, 3) or img.dtype != torch.float32 or not img.is_contiguous():
            raise ValueError("image must be a contiguous float32 [%d,%d,%d,3] tensor" % (B, h, w))
        stream = torch.cuda.current_stream(img.device).cuda_stream
        check(_lib.load().cp_pad_channels_3to4(img.data_ptr(), self.img4.data_ptr(), B * h * w, stream), "cp_pad_channels_3to4")
        for step in self.steps[:self.encoder_steps]:
            step(stream)
        return [self.taps[n] for n in ("x2s", "x4s", "x8s", "x16s", "x32s")]

    def _whole_records(self, L, K, pixels, new):
        """Both fused heads write slices of the same [pixels][K + V] output records -- 36 and 108 of every 144 bytes, in launches far apart -- and a
        partly written 128-byte line costs the memory system a read-modify-write (measured on the training heads: 2.2-2.7x the time of the same bytes
        as whole lines; tools/debug/head_probe.py).  So block 5's head writes DENSE rows of K logits (self.seg_dense) and block 10's, the last
        writer, copies them in front of its own columns (cp_conv_desc.head_prefix, the HS_PREFIX instantiations of csrc/conv_hsplit.hip): whole lines
        only.  Where it applies: both heads fused on the 2-byte-pipe kernel, block 10 a partial convolution, 8 <= K <= 12, K + V a multiple of 4.
        Opt-in (CASAPOSE_INFER_HEAD_RECORDS=1): see WHOLE_RECORDS."""
        self.seg_dense = None
        b5, b10 = L.get("pv_block_5_conv2d"), L.get("pv_block_10_prepare_conv2d")
        if (not WHOLE_RECORDS or not (self.labels_from_head and self.fuse_head2) or b5 is None or b10 is None or not (8 <= K <= 12) or self.out_ld % 4
                or b5.split_mode not in (3, _lib.PLANES_F16X2) or b10.split_mode not in (3, _lib.PLANES_F16X2) or not b10.desc.tap_label
                or b10.desc.head_cout + K != self.out_ld):
            return
        bound = {(id(l), f): o for l, o, f in self._out_bound}
        if bound.get((id(b5), "head_out")) != 0 or bound.get((id(b10), "head_out")) != K:
            return
        self.seg_dense = new(pixels, K)
        b5.desc.head_out, b5.desc.head_out_ld = self.seg_dense.data_ptr(), K
        b10.desc.head_prefix, b10.desc.head_prefix_n, b10.desc.head_prefix_ld = self.seg_dense.data_ptr(), K, K
        self._out_bound = [(l, o, f) for l, o, f in self._out_bound if not (l is b5 and f == "head_out") and not (l is b10 and f == "head_out")]
        self._out_bound.append((b10, 0, "head_out"))   # block 10 addresses the record itself

    def micro_steps(self, img: torch.Tensor, out: torch.Tensor):
        """The launches of run(img, out=out) with the estimated mask as a list of (tag, fn(stream)): "M" = matrix-pipe kernels (convolutions, the
        Winograd GEMMs), "H" = HBM-bound passes (Winograd transforms, pooling, label pyramid, resampling, channel padding).  Binds `out`."""
        lib = _lib.load()
        B, h, w = self.batch, self.h, self.w
        if tuple(img.shape) != (B, h, w, 3) or img.dtype != torch.float32 or not img.is_contiguous():
            raise ValueError("image must be a contiguous float32 [%d,%d,%d,3] tensor" % (B, h, w))
        if WINO_CHUNK or WINO_GROUPED_CONV:
            raise ValueError("the two-stream forward does not combine with CASAPOSE_WINO_CHUNK / CASAPOSE_WINO_GROUPED_CONV")
        self.out = out
        for layer, off, field in self._out_bound:
            setattr(layer.desc, field, out.data_ptr() + 4 * off)
        self.seg_input_ptr = None
        steps = [("H", lambda stream: check(lib.cp_pad_channels_3to4(img.data_ptr(), self.img4.data_ptr(), B * h * w, stream), "cp_pad_channels_3to4"))]
        for st in self.steps:
            owner = getattr(st, "__self__", None)
            if isinstance(owner, WinoConv):
                steps += owner.micro_steps()
            elif isinstance(owner, FusedConv):
                steps.append(("M", st))
            else:
                steps.append(("H", st))
        return steps

    def run(self, img: torch.Tensor, seg_input: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        lib = _lib.load()
        B, h, w = self.batch, self.h, self.w
        if tuple(img.shape) != (B, h, w, 3) or img.dtype != torch.float32 or not img.is_contiguous():
            raise ValueError("image must be a contiguous float32 [%d,%d,%d,3] tensor" % (B, h, w))
        stream = torch.cuda.current_stream(img.device).cuda_stream
        if out is None:
            out = torch.empty(B, h, w, self.out_ld, dtype=torch.float32, device=img.device)
        self.out = out
        for layer, off, field in self._out_bound:
            setattr(layer.desc, field, out.data_ptr() + 4 * off)
        if seg_input is not None:
            if tuple(seg_input.shape) != (B, h, w, self.net.seg_dim) or seg_input.dtype != torch.float32 or not seg_input.is_contiguous():
                raise ValueError("segmentation input must be a contiguous float32 [%d,%d,%d,%d] tensor" % (B, h, w, self.net.seg_dim))
            self.seg_input_ptr = seg_input.data_ptr()
        else:
            self.seg_input_ptr = None
        check(lib.cp_pad_channels_3to4(img.data_ptr(), self.img4.data_ptr(), B * h * w, stream), "cp_pad_channels_3to4")
        guard = self.net.conv_planes == _lib.PLANES_F16X2 and self.net.f16x2_guard
        if guard and not self.needs_calibration:
            self._poll_monitor()   # (may re-arm the calibration: a finished armed forward found a layer outside the band)
        if self.needs_calibration:
            self._calibrate(stream)
        elif guard and F16X2_MONITOR:
            self._run_armed(stream, fresh=False)   # sticky: the slots keep the maxima of every forward since they were last read
            self._since_monitor += 1
            if self._since_monitor >= F16X2_MONITOR_EVERY and self._mon_event is None:
                self._mon_host.copy_(self._mon, non_blocking=True)   # judged at the start of a later forward, when the copy has landed
                self._mon.zero_()                                    # (stream-ordered behind the copy)
                self._mon_event = torch.cuda.Event()
                self._mon_event.record(torch.cuda.current_stream(img.device))
                self._since_monitor = 0
        else:
            for step in self.steps:
                step(stream)
            self._since_monitor += 1
        if self.net.pvnet:
            _LABEL_CACHE.pop(out.untyped_storage().data_ptr(), None)
            return out
        if seg_input is None:  # labels[0] is the arg-max of THIS output's logits
            _remember_labels(out, self.labels[0].clone())
        else:
            _LABEL_CACHE.pop(out.untyped_storage().data_ptr(), None)
        return out


class CasaposeNet:
    """Parameters + launch plans of casapose_c_gcu5 on one GPU."""

    def __init__(self, params: Dict[str, np.ndarray], seg_dim: int, ver_dim: int, device: torch.device,
                 decoder_dims: Sequence[int] = DECODER_DIMS_DEFAULT, fuse_upsample: bool = True, fuse_heads: bool = True,
                 partial: Sequence[bool] = PARTIAL_DEFAULT, guided: Sequence[bool] = GUIDED_DEFAULT, use_winograd: bool = True,
                 bilinear: Sequence[bool] = BILINEAR_DEFAULT, pvnet: bool = False, shared: Sequence[bool] = (False,) * 5,
                 reuse_first: bool = False, skips2: bool = True, conv_mode: Optional[str] = None, f16x2_guard: Optional[bool] = None):
        _lib.load()  # fail loudly if the HIP library is missing
        if device.type != "cuda":
            raise _lib.CasaposeHipError("casapose_amd runs on a ROCm GPU only (got device %s); there is no CPU fallback" % device)
        self.device = device
        self.seg_dim, self.ver_dim = seg_dim, ver_dim
        self.decoder_dims = tuple(decoder_dims)
        self.partial, self.guided = tuple(bool(v) for v in partial), tuple(bool(v) for v in guided)
        self.bilinear = tuple(bool(v) for v in bilinear)
        self.pvnet = bool(pvnet)  # PVNet (pose_models.py:645-696): decoder 1 only, ONE 1x1 head producing seg + vertex channels
        # weight sharing between the decoders (pose_models.py:699-1362, the `_sw*` registry entries): shared[i] -- blocks i+1 and i+6
        # use the PartialConvolution weights pv_block_{i+1}_{i+6}_conv2d; reuse_first -- block 6 normalises the raw output of block
        # 1's convolution instead of convolving; skips2 False -- decoder 2 has no skip connections
        self.shared = tuple(bool(v) for v in shared)
        self.reuse_first, self.skips2 = bool(reuse_first), bool(skips2)
        self.fuse_upsample = fuse_upsample
        self.fuse_heads = fuse_heads
        self.use_winograd = use_winograd and os.environ.get("CASAPOSE_NO_WINOGRAD", "0") != "1"
        mode = conv_mode if conv_mode is not None else os.environ.get("CASAPOSE_INFER_CONV_MODE", DEFAULT_INFER_CONV_MODE)
        if mode not in ("f32", "split", "f16x2", "bf16"):
            raise ValueError("conv_mode must be f32, split, f16x2 or bf16 (got %r)" % mode)
        self.conv_mode = mode
        self.conv_planes = {"f32": 0, "split": 3, "f16x2": _lib.PLANES_F16X2, "bf16": 1}[mode]
        self.f16x2_guard = F16X2_GUARD if f16x2_guard is None else bool(f16x2_guard)
        self.f16x2_fallback: Dict[str, str] = {}   # layer -> why it left f16x2 (ForwardPlan._calibrate); reset by set_params
        self._f16x2_warned = False
        self.plans: Dict[Tuple[int, int, int], ForwardPlan] = {}
        self._twin: Optional["CasaposeNet"] = None   # the second half-batch's layer objects (two-stream forward)
        self._streams = None
        self.set_params(params)

    def recalibrate(self):
        """Forget everything the f16x2 range guard has decided -- demoted layers (f16x2_fallback), powers of two on V and on the heads' operands -- and
        calibrate again on the next forward.  For a caller whose first batch was not representative (a blank warm-up image); the monitor does the
        incremental form of this by itself when a later batch leaves the band."""
        self.f16x2_fallback, self._f16x2_warned = {}, False
        self.plans.clear()
        if self._twin is not None:
            self._twin.recalibrate()

    def planes_for(self, layer_name: str) -> int:
        """operand planes of one layer: the conv mode's, or 3 (exact split) once the f16x2 guard has demoted the layer"""
        return 3 if (self.conv_planes == _lib.PLANES_F16X2 and layer_name in self.f16x2_fallback) else self.conv_planes

    def set_params(self, params: Dict[str, np.ndarray]):
        self.params = {k: np.asarray(v, dtype=np.float32) for k, v in params.items()}
        self.f16x2_fallback, self._f16x2_warned = {}, False   # new parameters: every plan calibrates again on its first forward
        if getattr(self, "_twin", None) is not None:
            self._twin.set_params(params)
        dev = self.device
        p = self.params
        tabs: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

        def put(name, pair):
            tabs[name] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in pair)

        put("bn_data", fold_bn(p, "bn_data", pad_to=4))
        put("bn0", fold_bn(p, "bn0"))
        put("bn1", fold_bn(p, "bn1"))
        for s in range(4):
            for u in range(2):
                base = "stage%d_unit%d_" % (s + 1, u + 1)
                put(base + "bn1", fold_bn(p, base + "bn1"))
                put(base + "bn2", fold_bn(p, base + "bn2"))
        for i in range(5):
            put("pv_block_%d_bn" % (i + 1), fold_bn(p, "pv_block_%d_bn" % (i + 1)))
            if not self.pvnet:
                put("pv_block_%d_clade" % (i + 6), fold_clade(p, "pv_block_%d_clade" % (i + 6)))
        self.device_tables = tabs

        L: Dict[str, FusedConv] = {}
        Wn: Dict[str, WinoConv] = {}

        def add(name, key, layout, k, cout, sources, stride=1, dil=1, pad=None, partial=False):
            L[name] = FusedConv(name, p[key], layout, k, k, cout, sources, dev, want_split=bool(self.conv_planes))
            pad = dil * (k // 2) if pad is None else pad
            planes = 2 if self.conv_mode == "bf16" else (self.conv_planes if self.conv_mode in ("split", "f16x2") and not WINO_GEMM_F32 else None)
            if self.use_winograd and not partial and wino_eligible(k, stride, dil, pad, sources, cout, split_gemm=bool(planes) or WINO_GEMM_SPLIT, f16x2=self.conv_mode == "f16x2"):
                Wn[name] = WinoConv(name, p[key] if layout == 0 else np.transpose(p[key], (1, 2, 0, 3)), cout, sources, dev, split_planes=planes)

        add("conv0", "conv0.kernel", 0, 7, 64, [(4, 3)])
        cin = 64
        for s, f in enumerate(STAGE_FILTERS):
            for u in range(2):
                base = "stage%d_unit%d_" % (s + 1, u + 1)
                st_, dl_ = (STAGE_STRIDE[s] if u == 0 else 1), STAGE_DILATION[s]
                if u == 0:
                    add(base + "sc", base + "sc.kernel", 0, 1, f, [(cin, cin)], stride=st_)
                add(base + "conv1", base + "conv1.kernel", 0, 3, f, [(cin, cin)], stride=st_, dil=dl_)
                add(base + "conv2", base + "conv2.kernel", 0, 3, f, [(f, f)], dil=dl_)
                cin = f
        dims = self.decoder_dims
        skip_c = [None, (128, 128), (64, 64), (64, 64), (4, 3)]
        for i in range(5):
            srcs = [(512, 512)] if i == 0 else [(dims[i - 1], dims[i - 1]), skip_c[i]]
            shared_key = "pv_block_%d_%d_conv2d.weights" % (i + 1, i + 6)
            if self.shared[i]:  # one-input PartialConvolution = ordinary SAME conv with [Cin,3,3,Cout] weights
                add("pv_block_%d_conv2d" % (i + 1), shared_key, 1, 3, dims[i], srcs)
            else:
                add("pv_block_%d_conv2d" % (i + 1), "pv_block_%d_conv2d.kernel" % (i + 1), 0, 3, dims[i], srcs)
            if self.pvnet or (i == 0 and self.reuse_first):
                continue
            srcs2 = srcs if (self.skips2 or i == 0) else [(dims[i - 1], dims[i - 1])]
            name2 = ("pv_block_%d_prepare_conv2d" if self.partial[i] else "pv_block_%d_conv2d") % (i + 6)
            if self.shared[i]:
                add(name2, shared_key, 1, 3, dims[i], srcs2, partial=self.partial[i])
            elif self.partial[i]:
                add(name2, "pv_block_%d_prepare_conv2d.weights" % (i + 6), 1, 3, dims[i], srcs2, partial=True)
            else:
                add(name2, "pv_block_%d_conv2d.kernel" % (i + 6), 0, 3, dims[i], srcs2)
        if self.pvnet:
            add("pv_final_conv", "pv_final_conv.kernel", 0, 1, self.seg_dim + self.ver_dim, [(dims[4], dims[4])])
        else:
            add("pv_final_conv_segmentation", "pv_final_conv_segmentation.kernel", 0, 1, self.seg_dim, [(dims[4], dims[4])])
            add("pv_final_conv_vertex", "pv_final_conv_vertex.kernel", 0, 1, self.ver_dim, [(dims[4], dims[4])])
        self.layers_by_name = L
        self.wino_by_name = Wn
        self.plans.clear()

    def plan(self, batch: int, h: int, w: int) -> ForwardPlan:
        key = (batch, h, w)
        if key not in self.plans:
            # descriptors live inside the FusedConv objects, so one plan is active at a time
            self.plans.clear()
            self.plans[key] = ForwardPlan(self, batch, h, w, self.fuse_upsample, self.fuse_heads)
        return self.plans[key]

    def forward(self, img: torch.Tensor, seg_input: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        b, h, w, _ = img.shape
        if TWO_STREAM and seg_input is None and not self.pvnet and b >= 2 and b % 2 == 0 and not (WINO_CHUNK or WINO_GROUPED_CONV):
            return self._forward_two_streams(img, out)
        return self.plan(b, h, w).run(img, seg_input, out)

    def _forward_two_streams(self, img: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
        """The batch as two halves software-pipelined over TWO HIP streams (round 4): matrix-pipe kernels ("M": convolutions, Winograd GEMMs)
        on one stream, HBM-bound passes ("H": Winograd transforms, pooling, label pyramid, resampling) on the other, events where a half's
        chain changes stream.  The persistent "M" kernels are launched with TWO_STREAM_BLOCKS (224 of 256) blocks, so the "H" kernels of one
        half find free CUs while the other half's convolution runs: measured on the stage-4 shapes GEMM 0.325 ms + transforms 0.121 ms alone,
        0.364 ms together.  Each half has its own CasaposeNet (descriptors and buffers live in the layer objects; the weights
        are duplicated: 59 MB); results are those of two independent half-batch forwards -- the network has no cross-image term."""
        b, h, w, _ = img.shape
        hb = b // 2
        if self._twin is None:
            self._twin = CasaposeNet(self.params, self.seg_dim, self.ver_dim, self.device, decoder_dims=self.decoder_dims, fuse_upsample=self.fuse_upsample,
                                     fuse_heads=self.fuse_heads, partial=self.partial, guided=self.guided, use_winograd=self.use_winograd,
                                     bilinear=self.bilinear, pvnet=False, shared=self.shared, reuse_first=self.reuse_first, skips2=self.skips2,
                                     conv_mode=self.conv_mode, f16x2_guard=self.f16x2_guard)
            self._streams = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
        if out is None:
            out = torch.empty(b, h, w, self.seg_dim + self.ver_dim, dtype=torch.float32, device=img.device)
        plans = [self.plan(hb, h, w), self._twin.plan(hb, h, w)]
        for i, pl in enumerate(plans):   # f16x2 range guard: a plan's first forward runs layer by layer on the current stream
            if pl.needs_calibration:
                pl.run(img[i * hb:(i + 1) * hb], out=out[i * hb:(i + 1) * hb])
        seqs = [plans[0].micro_steps(img[:hb], out[:hb]), plans[1].micro_steps(img[hb:], out[hb:])]
        lib = _lib.load()
        cur = torch.cuda.current_stream(self.device)
        streams = {"M": self._streams[0], "H": self._streams[1]}
        for s_ in streams.values():
            s_.wait_stream(cur)
        old_blocks = lib.cp_get_persistent_blocks()
        check(lib.cp_set_persistent_blocks(min(old_blocks, TWO_STREAM_BLOCKS)), "cp_set_persistent_blocks")
        try:
            pos, last = [0, 0], [None, None]   # next micro-step / (tag, event) of the last issued step per half
            lead = int(TWO_STREAM_SKEW * len(seqs[0]))
            if TWO_STREAM_MODE == "half":
                # one in-order stream per HALF (no cross-stream waits inside a chain): the hardware runs whatever of the two queues fits;
                # half 1 starts when half 0 has passed `lead` steps, so that its transform-heavy encoder meets half 0's convolution-only decoders
                s0, s1 = self._streams
                gate = None
                for k_, (tag, fn) in enumerate(seqs[0]):
                    fn(s0.cuda_stream)
                    if k_ + 1 == lead:
                        gate = torch.cuda.Event()
                        gate.record(s0)
                    if k_ + 1 >= lead and pos[1] < len(seqs[1]):     # keep both queues fed: one step of half 1 per step of half 0
                        if pos[1] == 0 and gate is not None:
                            s1.wait_event(gate)
                        seqs[1][pos[1]][1](s1.cuda_stream)
                        pos[1] += 1
                while pos[1] < len(seqs[1]):
                    seqs[1][pos[1]][1](s1.cuda_stream)
                    pos[1] += 1
                pos[0] = len(seqs[0])

            def issue(k):
                tag, fn = seqs[k][pos[k]]
                st = streams[tag]
                if last[k] is not None and last[k][0] != tag:
                    st.wait_event(last[k][1])                   # this half's previous step ran on the other stream
                fn(st.cuda_stream)
                pos[k] += 1
                ev = None
                if pos[k] < len(seqs[k]) and seqs[k][pos[k]][0] != tag:   # the next step of this half changes stream: it will wait for this one
                    ev = torch.cuda.Event()
                    ev.record(st)
                last[k] = (tag, ev)

            # "tag" mode: matrix-pipe kernels on one stream, HBM-bound passes on the other, events where a half's chain changes stream; half 0
            # runs ahead by `lead` steps, then the two lists are issued alternately -- the streams serialise equal tags in issue order
            for _ in range(lead if pos[0] == 0 else 0):
                issue(0)
            while pos[0] < len(seqs[0]) or pos[1] < len(seqs[1]):
                for k in (1, 0):
                    if pos[k] < len(seqs[k]):
                        issue(k)
        finally:
            check(lib.cp_set_persistent_blocks(old_blocks), "cp_set_persistent_blocks")
        for s_ in streams.values():
            cur.wait_stream(s_)
        _remember_labels(out, torch.cat([plans[0].labels[0], plans[1].labels[0]]))
        return out
