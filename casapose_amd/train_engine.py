"""Training step of casapose_c_gcu5 on MI355X: forward with batch statistics, losses, hand-written
backward, Adam -- the path of train_casapose.py:494-611 (runnetwork/train_step) -- issued as a static
launch plan over libcasapose_hip.so.

Design
  * one flat fp32 MASTER parameter buffer (Keras layout per variable, names of SURVEY Appendix A) with a
    flat gradient and the two Adam moments beside it: one Adam launch and one all-reduce per step;
  * every kernel layout of a convolution (forward K order, halo fragment order, the flipped/transposed
    data-gradient packs) is a device GATHER of the master buffer through an index map built once by pushing
    an index ramp through the same host packers the inference engine uses; the packed weight gradient of
    cp_conv2d_wgrad_f32 goes back through the same map (scatter);
  * the plan is a tape: a list of ops with forward()/backward(); tensors with several consumers
    accumulate gradients in place (first writer overwrites, later writers add -- for convolution
    data-gradients through the kernel's fused residual input);
  * SyncBatchNormalization: the fp64 sum tables of cp_bn_stats_f32 / cp_bn_act_bwd_reduce_f32 are
    all-reduced across replicas when a process group is given (parallel.py), nothing else is exchanged in
    the forward/backward; the flat gradient is SUM-all-reduced before Adam (MirroredStrategy semantics,
    train_casapose.py:641-643).

The decoder-2 conditioning is the hard label map (arg-max of the logits or, with
train_vectors_with_ground_truth, the ground-truth segmentation: train_casapose.py:522-524) and is a constant
of the gradient, exactly like the saturated softmax of the reference (pose_models.py:547-552) whose
derivative vanishes in fp32.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, parallel
from ._lib import ConvDesc, check
from .engine import BILINEAR_DEFAULT, BN_EPS, DECODER_DIMS_DEFAULT, GUIDED_DEFAULT, PARTIAL_DEFAULT, STAGE_DILATION, STAGE_FILTERS, STAGE_STRIDE

BN_MOMENTUM = 0.99  # resnet.py:43 (Keras default elsewhere)


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# parameters
# ------------------------------------------------------------------------------------------------
def forward_order() -> List[str]:
    """Layer names in the order the forward uses them (resnet.py:246-305; pose_models.py:541-616).  The flat parameter buffer follows
    this order, so the backward completes it from the END towards the start and gradient buckets are contiguous tail slices."""
    order = ["bn_data", "conv0", "bn0"]
    for s in range(1, 5):
        for u in range(1, 3):
            base = "stage%d_unit%d_" % (s, u)
            order += [base + "bn1", base + "sc", base + "conv1", base + "bn2", base + "conv2"]
    order.append("bn1")
    for i in range(1, 6):
        order += ["pv_block_%d_conv2d" % i, "pv_block_%d_%d_conv2d" % (i, i + 5), "pv_block_%d_bn" % i]
    order += ["pv_final_conv_segmentation", "pv_final_conv"]
    for i in range(6, 11):
        order += ["pv_block_%d_prepare_conv2d" % i, "pv_block_%d_conv2d" % i, "pv_block_%d_clade" % i]
    order.append("pv_final_conv_vertex")
    return order


# gradient buckets = groups of consecutive layers whose all-reduce is launched as soon as the backward has passed them
BUCKET_STARTS = ("bn_data", "stage4_unit1_bn1", "pv_block_1_conv2d", "pv_block_6_prepare_conv2d")


# refreshes between two evaluations of max |w| for the f16x2 weight scale (a host synchronisation per layer).  16: an Adam step moves a weight by about
# the learning rate (<= 1e-3 here), so over 16 steps max |w| of an initialised layer (he_uniform: >= 0.036) cannot leave the 16x headroom; round 4's 256 could
F16X2_RESCALE_EVERY = int(os.environ.get("CASAPOSE_F16X2_RESCALE_EVERY", "16"))


def train_fwd_f16x2() -> bool:
    """The FORWARD convolutions of the training plan in the fp16 two-way split (csrc/split_f16.h: three products, fp32-level error) wherever the conv
    mode is "split"; CASAPOSE_TRAIN_FWD=split keeps the exact three-way bf16 split there too.  Only the forward: its operands are normalised
    activations and weights (scaled by a power of two); the backward's operands are gradients of arbitrary magnitude and stay on the exact bf16 split.
    Rounds 4 / 5 kept this an opt-in (57.2 against 53.4 ms per step) for two reasons.  (1) Accuracy: the exact split is BETTER than fp32 (operands
    exact) -- on the one ill-conditioned gradient comparison of the suite (config 13: the proxy-voting loss divides by |v|^2) the worst per-variable
    error is 6e-5 with the exact forward and 4.4e-3 with this one, which is what the fp32 MFMA gives as well
    (tests/test_gpu_train.py::test_config13_bpnp_step_at_448_matches_autograd): fp32-LEVEL, i.e. the reference's arithmetic.  (2) Nobody watched the
    fp16 range condition on activations that change with every step.  Round 6 removed (2): every f16x2 forward launch of the plan reports max |x| of what
    it converts into a device-side monitor slot (cp_f16x2_monitor_set, include/casapose_hip.h), the slots are judged every F16X2_TRAIN_CHECK_EVERY
    steps without a synchronisation, and an op whose operands have left the band moves its forward to the exact split for good (TrainPlan._poll_f16x2).
    With that the fp32-level forward is the DEFAULT."""
    return os.environ.get("CASAPOSE_TRAIN_FWD", "f16x2") == "f16x2"


def train_bwd_f16x2() -> bool:
    """The Winograd DATA-GRADIENT GEMMs of the training plan in the fp16 two-way split as well (round 6; CASAPOSE_TRAIN_BWD=split keeps the exact
    three-way bf16 split).  A gradient has no natural magnitude, so the operand gets one: the input transform of dY multiplies by 2^e (its
    per-channel affine with a constant table: exact), the GEMM's accumulator factor takes 2^-e.  e comes from the monitor slot the transform
    reports max |V| into: the FIRST backward of a plan runs on the exact split with the slots armed and is followed by one synchronous reading
    (like the inference plan's calibration); from then on the slots are judged with the forward's, every F16X2_TRAIN_CHECK_EVERY steps without a
    synchronisation, and e follows when the maximum has drifted out of [2^7, 2^13] (target [2^10, 2^11): 8x headroom to the band's end, 32x to
    fp16's clamp).  A non-finite maximum returns the op to the exact split.
    The DIRECT 3x3 layers' data gradients (csrc/conv_hsplit.hip converts its source as it is) get their magnitude from ONE power of two on the
    loss instead (TrainPlan.loss_exp: the loss weights are multiplied by 2^E, every gradient of the backward carries the factor, the flat
    gradient is multiplied by 2^-E before anything reads it -- all exact): E puts the LARGEST max |dY| of those layers at [2^10, 2^11), the
    layers whose own maximum then sits inside [1, 2^13] run on the fp16 pair (measured spread between the layers of this network: 2^9 - 2^12.5,
    tools/debug/grad_ranges.py), the others stay on the exact split.  max |dY| comes from cp_amax_f32 passes on the check steps only."""
    return os.environ.get("CASAPOSE_TRAIN_BWD", "f16x2") == "f16x2"


# steps between two readings of the forward range monitor (asynchronous copy to pinned memory, judged at the start of a later step)
F16X2_TRAIN_CHECK_EVERY = max(1, int(os.environ.get("CASAPOSE_F16X2_TRAIN_CHECK_EVERY", "16")))


def f16x2_scale(cache: dict, weights: torch.Tensor) -> float:
    """the power of two the weights are multiplied by before their fp16 split (cp_f16x2_weight_scale: max |w| -> [2^11, 2^12), i.e. 16x headroom to
    the fp16 range).  It needs max |w| on the host, so it is re-evaluated every F16X2_RESCALE_EVERY refreshes, not every step."""
    n = cache.get("n", 0)
    cache["n"] = n + 1
    if "scale" not in cache or n % F16X2_RESCALE_EVERY == 0 or cache.get("max", 0.0) == 0.0:   # (an all-zero tensor has no scale yet: look again next time)
        cache["max"] = float(weights.abs().max())
        cache["scale"] = float(_lib.load().cp_f16x2_weight_scale(cache["max"]))
    return cache["scale"]


def conv_split_planes() -> int:
    """bf16-pipe mode of the shallow 3x3 convolutions of the TRAINING plan (csrc/conv_hsplit.hip), read from CASAPOSE_CONV_MODE:
    "split" (default) = 3 planes, exact three-way bf16 split, fp32-equivalent; "bf16" = 1 plane, operands rounded to bf16 (BASELINE configs[2]);
    "f32" = 0, the fp32-MFMA kernels."""
    mode = os.environ.get("CASAPOSE_CONV_MODE", "split")
    if mode not in ("split", "bf16", "f32"):
        raise ValueError("CASAPOSE_CONV_MODE must be split, bf16 or f32 (got %r)" % mode)
    return {"split": 3, "bf16": 1, "f32": 0}[mode]


class ParamStore:
    """Flat master parameters / gradients / Adam moments with named views."""

    def __init__(self, params: Dict[str, np.ndarray], device: torch.device):
        self.device = device
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.state: Dict[str, torch.Tensor] = {}  # non-trainable: moving statistics
        chunks, off = [], 0
        rank = {n: i for i, n in enumerate(forward_order())}
        ordered = sorted(params.items(), key=lambda kv: rank.get(kv[0].split(".")[0], len(rank)))  # stable: unknown layers keep their order at the end
        for name, v in ordered:
            a = np.asarray(v, dtype=np.float32)
            if name.endswith(".moving_mean") or name.endswith(".moving_variance"):
                self.state[name] = torch.from_numpy(a.copy()).to(device)
                continue
            n = a.size
            self.offsets[name] = (off, tuple(a.shape))
            chunks.append(a.reshape(-1))
            pad = (-n) % 4  # keep every variable 16-byte aligned
            if pad:
                chunks.append(np.zeros(pad, np.float32))
            off += n + pad
        self.size = off
        self.theta = torch.from_numpy(np.concatenate(chunks)).to(device)
        self.grad = torch.zeros_like(self.theta)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.step_count = 0

    # ---- packed-weight arena: every kernel layout of every layer lives in ONE buffer refreshed by ONE gather of the master parameters
    def pack_reset(self):
        self._pack_parts: List[np.ndarray] = []
        self._pack_used = 0
        self._pack_cap = 8 * self.size
        self.pack_arena = torch.empty(self._pack_cap, dtype=torch.float32, device=self.device)
        self.pack_idx: Optional[torch.Tensor] = None

    def pack_alloc(self, idx_rel: np.ndarray, master_off: int) -> torch.Tensor:
        """A packed-layout buffer of len(idx_rel) floats whose element i is theta[master_off + idx_rel[i]] (0 where idx_rel[i] < 0)."""
        n = int(idx_rel.size)
        if getattr(self, "_pack_parts", None) is None or self.pack_idx is not None:
            self.pack_reset()
        npad = n + ((-n) % 4)
        if self._pack_used + npad > self._pack_cap:
            raise RuntimeError("packed-weight arena exhausted (%d + %d > %d floats)" % (self._pack_used, npad, self._pack_cap))
        part = np.full(npad, -1, np.int32)
        part[:n] = np.where(idx_rel >= 0, idx_rel.astype(np.int64) + master_off, -1).astype(np.int32)
        self._pack_parts.append(part)
        view = self.pack_arena[self._pack_used:self._pack_used + n]
        self._pack_used += npad
        return view

    def pack_finalize(self):
        self.pack_idx = torch.from_numpy(np.concatenate(self._pack_parts)).to(self.device) if self._pack_parts else None

    def pack_refresh(self, stream: int):
        """arena[i] = theta[pack_idx[i]]: all layers' kernel layouts in one launch"""
        if self.pack_idx is not None:
            check(_lib.load().cp_gather_f32(self.theta.data_ptr(), self.pack_idx.data_ptr(), self.pack_idx.numel(), self.pack_arena.data_ptr(), stream),
                  "cp_gather_f32(arena)")

    def view(self, name: str, of: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shape = self.offsets[name]
        n = int(np.prod(shape))
        return (self.theta if of is None else of)[off:off + n].view(shape)

    def grad_view(self, name: str) -> torch.Tensor:
        return self.view(name, self.grad)

    def export(self) -> Dict[str, np.ndarray]:
        out = {k: self.view(k).detach().cpu().numpy().copy() for k in self.offsets}
        out.update({k: v.detach().cpu().numpy().copy() for k, v in self.state.items()})
        return out

    def adam_step(self, lr: float, stream: int, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        self.step_count += 1
        check(_lib.load().cp_adam_step_f32(self.theta.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.size,
                                           lr, beta1, beta2, eps, self.step_count, grad_scale, stream), "cp_adam_step_f32")


class TT:
    """Activation tensor [B,H,W,C] (contiguous) with an optional gradient buffer."""

    __slots__ = ("data", "grad", "has_grad", "needs_grad", "name")

    def __init__(self, data: torch.Tensor, needs_grad: bool = True, name: str = ""):
        self.data = data
        self.needs_grad = needs_grad
        self.grad = torch.empty_like(data) if needs_grad else None
        self.has_grad = False
        self.name = name

    @property
    def c(self):
        return self.data.shape[-1]

    @property
    def pixels(self):
        return self.data.numel() // self.data.shape[-1]


def _index_map(pack: Callable[[np.ndarray, np.ndarray], None], ramp_hwio: np.ndarray, n_out: int) -> np.ndarray:
    """Run a host packer over (index+1) stored as fp32 and recover the int32 gather map (-1 = zero fill)."""
    assert ramp_hwio.max() < (1 << 24)
    src = np.ascontiguousarray(ramp_hwio + 1, dtype=np.float32)
    dst = np.empty(n_out, dtype=np.float32)
    pack(src, dst)
    return (dst.astype(np.int64) - 1).astype(np.int32)


class TrainConv:
    """One convolution layer of the training plan: packs, descriptors, forward / wgrad / dgrad launches."""

    def __init__(self, store: ParamStore, key: str, layout: int, k: int, cout: int, sources: Sequence[Tuple[int, int]],
                 grad_sources: Sequence[bool]):
        lib = _lib.load()
        dev = store.device
        self.store, self.key, self.k, self.cout = store, key, k, cout
        self.sources = list(sources)
        self.name = key.rsplit(".", 1)[0]
        off, shape = store.offsets[key]
        n = int(np.prod(shape))
        self.master = store.theta[off:off + n]
        self.master_grad = store.grad[off:off + n]
        ramp = np.arange(n, dtype=np.int64).reshape(shape)
        hwio = ramp if layout == 0 else ramp.transpose(1, 2, 0, 3)  # [kh,kw,cin,cout] view of the master indices
        ns = len(sources)
        chans = (C.c_int * 2)(*([s[0] for s in sources] + [0] * (2 - ns)))
        real = (C.c_int * 2)(*([s[1] for s in sources] + [0] * (2 - ns)))
        self.ktot = lib.cp_conv_ktot(k, k, ns, chans)

        def pack_fwd(src, dst):
            check(lib.cp_conv_pack_weights_host(src.ctypes.data, 0, k, k, cout, ns, chans, real, dst.ctypes.data), "pack " + key)

        self._off = off
        imap = _index_map(pack_fwd, hwio, cout * self.ktot)
        self.idx_fwd = torch.from_numpy(imap).to(dev)
        self.wp = store.pack_alloc(imap, off)
        self.dwp = torch.empty(cout * self.ktot, dtype=torch.float32, device=dev)
        self.idx_halo = self.wp_halo = None
        if k == 3 and cout <= 64 and sources[0][0] % 32 == 0 and (ns == 1 or sources[1][0] == 4 or sources[1][0] % 32 == 0):
            nfl = lib.cp_conv_halo_weight_floats(cout, ns, chans)

            def pack_halo(src, dst):
                check(lib.cp_conv_pack_weights_halo_host(src.ctypes.data, 0, cout, ns, chans, real, dst.ctypes.data), "pack halo " + key)

            imap = _index_map(pack_halo, hwio, nfl)
            self.idx_halo = torch.from_numpy(imap).to(dev)
            self.wp_halo = store.pack_alloc(imap, off)
        elif k == 7 and cout == 64 and ns == 1 and sources[0][0] == 4:  # the stem (csrc/conv_stem.hip): its packing travels in weights_halo
            nfl = 25 * 2 * 64 * 4

            def pack_stem(src, dst):
                check(lib.cp_conv_pack_weights_stem_host(src.ctypes.data, 0, sources[0][1], dst.ctypes.data), "pack stem " + key)

            imap = _index_map(pack_stem, hwio, nfl)
            self.idx_halo = torch.from_numpy(imap).to(dev)
            self.wp_halo = store.pack_alloc(imap, off)
        # bf16-pipe kernel (csrc/conv_hsplit.hip) for the shallow 3x3 layers: fp32 image of its fragment stream in the arena, bf16 planes beside it
        self.split = None
        planes = self.mode_planes = conv_split_planes()   # read ONCE per layer: the backward and the accounting reuse it (bench.py changes the variable between plans)
        self.fwd_f16x2 = planes == 3 and train_fwd_f16x2()   # forward launches of this layer in the fp16 two-way split
        fwd_np = _lib.PLANES_F16X2 if self.fwd_f16x2 else planes
        deep_dgrad = False
        if layout == 0 and sum(s_[0] for s_ in sources) >= 256 and cout >= 128:
            # bf16 conv mode: the DATA gradient of these layers runs on the direct bf16-operand kernel (csrc/conv_bf16d.hip; 1.2-1.6x the two-plane
            # Winograd path at bs 32) -- the forward stays Winograd: its transformed input V is what the weight gradient multiplies, its output
            # transform accumulates the batch statistics and its input transform applies the normalisation (CASAPOSE_BF16_DEEP=0: Winograd everywhere)
            deep_dgrad = planes == 1 and k == 3 and os.environ.get("CASAPOSE_BF16_DEEP", "1") != "0"
            planes = 0   # the deep ordinary 3x3 layers run as Winograd (engine.wino_eligible); partial convolutions (layout 1) of that size do not
        if planes and k == 3 and cout <= 512 and cout % 4 == 0 and sources[0][0] % 16 == 0 and sources[0][0] != 4 and (
                ns == 1 or (sources[1][0] == 4 and cout <= 32) or sources[1][0] % 16 == 0):
            nfl = lib.cp_conv_split_weight_floats(cout, ns, chans)

            def pack_split(src, dst):
                check(lib.cp_conv_pack_weights_split_host(src.ctypes.data, 0, cout, ns, chans, real, dst.ctypes.data), "pack split " + key)

            imap = _index_map(pack_split, hwio, nfl)
            self.split = dict(f32=store.pack_alloc(imap, off), planes=torch.empty(nfl // 512 * (fwd_np & 15) * 1024, dtype=torch.uint8, device=dev), np=fwd_np,
                              idx=torch.from_numpy(imap).to(dev), descale=1.0, cache={})
        elif planes and k == 7 and cout == 64 and ns == 1 and sources[0][0] == 4 and os.environ.get("CASAPOSE_STEM_SPLIT", "1") != "0":
            # the stem on the bf16 matrix pipe (csrc/conv_stem_split.hip, round 4): same bookkeeping as the 3x3 layers -- fp32 image of the fragment
            # stream in the arena (re-gathered with every weight refresh), bf16 planes beside it; the weight gradient stays on the fp32 kernel
            nfl = lib.cp_conv_stem_split_weight_floats()

            def pack_stem_split(src, dst):
                check(lib.cp_conv_pack_weights_stem_split_host(src.ctypes.data, 0, sources[0][1], dst.ctypes.data), "pack stem split " + key)

            imap = _index_map(pack_stem_split, hwio, nfl)
            self.split = dict(f32=store.pack_alloc(imap, off), planes=torch.empty(nfl // 512 * (fwd_np & 15) * 1024, dtype=torch.uint8, device=dev), np=fwd_np,
                              idx=torch.from_numpy(imap).to(dev), stem=True, descale=1.0, cache={})
        # data-gradient packs: per source that needs a gradient, the flipped / transposed kernel
        self.dgrad: List[Optional[dict]] = []
        cpad = (cout + 31) // 32 * 32
        c0 = 0
        for s, (cs, cr) in enumerate(sources):
            if not grad_sources[s]:
                self.dgrad.append(None)
                c0 += cr
                continue
            sub = hwio[::-1, ::-1, c0:c0 + cr, :].transpose(0, 1, 3, 2)  # [kh,kw,cout,cr]: taps flipped, in/out swapped
            dch = (C.c_int * 2)(cpad, 0)
            dre = (C.c_int * 2)(cout, 0)
            kt = lib.cp_conv_ktot(k, k, 1, dch)

            def pack_d(src, dst, dch=dch, dre=dre, cr=cr):
                check(lib.cp_conv_pack_weights_host(src.ctypes.data, 0, k, k, cr, 1, dch, dre, dst.ctypes.data), "pack dgrad " + key)

            imap = _index_map(pack_d, np.ascontiguousarray(sub), cr * kt)
            ent = dict(idx=torch.from_numpy(imap).to(dev), w=store.pack_alloc(imap, off), cout=cr, cin=cpad, idx_halo=None, w_halo=None,
                       desc=ConvDesc())
            if k == 3 and cr <= 64:
                nfl = lib.cp_conv_halo_weight_floats(cr, 1, dch)

                def pack_dh(src, dst, dch=dch, dre=dre, cr=cr):
                    check(lib.cp_conv_pack_weights_halo_host(src.ctypes.data, 0, cr, 1, dch, dre, dst.ctypes.data), "pack dgrad halo " + key)

                imap = _index_map(pack_dh, np.ascontiguousarray(sub), nfl)
                ent["idx_halo"] = torch.from_numpy(imap).to(dev)
                ent["w_halo"] = store.pack_alloc(imap, off)
            ent["split"] = None
            ent["deep"] = deep_dgrad and cr % 128 == 0 and cr <= 512 and cpad % 16 == 0
            if (planes or ent["deep"]) and k == 3 and cr <= 512 and cr % 4 == 0:
                planes_d = planes or 1
                nfl = lib.cp_conv_split_weight_floats(cr, 1, dch)

                def pack_ds(src, dst, dch=dch, dre=dre, cr=cr):
                    check(lib.cp_conv_pack_weights_split_host(src.ctypes.data, 0, cr, 1, dch, dre, dst.ctypes.data), "pack dgrad split " + key)

                imap = _index_map(pack_ds, np.ascontiguousarray(sub), nfl)
                ent["split"] = dict(f32=store.pack_alloc(imap, off), planes=torch.empty(nfl // 512 * planes_d * 1024, dtype=torch.uint8, device=dev), np=planes_d,
                                    idx=torch.from_numpy(imap).to(dev), descale=1.0, cache={})
            self.dgrad.append(ent)
            c0 += cr
        self.desc = ConvDesc()
        self._keep: List = []
        self.layout = layout
        self.master_shape = tuple(shape)
        self.refresh_hooks: List[Callable[[int], None]] = []  # extra weight layouts owned by the ops (Winograd planes)
        if self.split is not None or any(e is not None and e["split"] is not None for e in self.dgrad):
            self.refresh_hooks.append(self._refresh_split)

    def _refresh_split(self, stream: int):
        """fp32 fragment images (just re-gathered from the master weights) -> bf16 planes of the bf16-pipe kernel"""
        lib = _lib.load()
        for sp in [self.split] + [e["split"] for e in self.dgrad if e is not None]:
            if sp is not None:
                scale = f16x2_scale(sp["cache"], sp["f32"]) if sp["np"] == _lib.PLANES_F16X2 else 1.0
                sp["descale"] = 1.0 / scale
                check(lib.cp_conv_split_weights_scaled_f32(sp["f32"].data_ptr(), sp["f32"].numel(), sp["np"], scale, sp["planes"].data_ptr(), stream),
                      "cp_conv_split_weights_scaled_f32")

    def refresh(self, stream: int, hooks_only: bool = False):
        """Re-pack the kernel layouts from the master weights (after an optimizer step / at start).  hooks_only: the plan refreshes the
        gather-type layouts of all layers with one launch over the packed-weight arena (ParamStore.pack_refresh)."""
        lib = _lib.load()
        m = self.master.data_ptr()
        if hooks_only:
            for hook in self.refresh_hooks:
                hook(stream)
            return
        for sp in [self.split] + [e["split"] for e in self.dgrad if e is not None]:   # fp32 fragment images of the bf16-pipe kernel, then their planes
            if sp is not None:
                check(lib.cp_gather_f32(m, sp["idx"].data_ptr(), sp["idx"].numel(), sp["f32"].data_ptr(), stream), "cp_gather_f32")
        for hook in self.refresh_hooks:
            hook(stream)
        check(lib.cp_gather_f32(m, self.idx_fwd.data_ptr(), self.idx_fwd.numel(), self.wp.data_ptr(), stream), "cp_gather_f32")
        if self.idx_halo is not None:
            check(lib.cp_gather_f32(m, self.idx_halo.data_ptr(), self.idx_halo.numel(), self.wp_halo.data_ptr(), stream), "cp_gather_f32")
        for ent in self.dgrad:
            if ent is None:
                continue
            check(lib.cp_gather_f32(m, ent["idx"].data_ptr(), ent["idx"].numel(), ent["w"].data_ptr(), stream), "cp_gather_f32")
            if ent["idx_halo"] is not None:
                check(lib.cp_gather_f32(m, ent["idx_halo"].data_ptr(), ent["idx_halo"].numel(), ent["w_halo"].data_ptr(), stream), "cp_gather_f32")


class ConvOp:
    """raw = conv(sources) [* row_scale] [+ residual]; backward: wgrad + per-source dgrad."""

    def __init__(self, layer: TrainConv, srcs: Sequence[Tuple[TT, int]], out_ptr_ld: Tuple[torch.Tensor, int, int], batch: int, in_h: int,
                 in_w: int, stride=1, dilation=1, pad=0, tap_label=None, row_scale=None, residual: Optional[TT] = None,
                 out: Optional[TT] = None, dy_ptr_ld: Optional[Tuple[torch.Tensor, int, int]] = None):
        """srcs: (tensor, ld).  out_ptr_ld = (storage tensor, float offset, ld) of the raw output.  dy_ptr_ld: where the
        gradient of the output lives (default: out.grad)."""
        self.layer, self.srcs, self.out, self.residual = layer, list(srcs), out, residual
        self.tap_label, self.row_scale = tap_label, row_scale
        self.accumulate_master = False  # True: another convolution on the same weights has already written the master gradient
        self.stride, self.dil, self.pad = stride, dilation, pad
        self.batch, self.in_h, self.in_w = batch, in_h, in_w
        k = layer.k
        eff = (k - 1) * dilation + 1
        self.out_h = (in_h + 2 * pad - eff) // stride + 1
        self.out_w = (in_w + 2 * pad - eff) // stride + 1
        d = layer.desc
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w = batch, in_h, in_w, self.out_h, self.out_w
        d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = layer.cout, k, k, stride, dilation, pad
        d.num_sources = len(srcs)
        for i, (t, ld) in enumerate(srcs):
            cs = d.src[i]
            cs.data, cs.channels, cs.ld, cs.mode = t.data.data_ptr(), layer.sources[i][0], ld, _lib.SRC_DIRECT
            cs.sel = cs.pre_scale = cs.pre_shift = None
        d.weights = layer.wp.data_ptr()
        d.weights_halo = _ptr(layer.wp_halo)
        d.tap_label, d.row_scale = _ptr(tap_label), _ptr(row_scale)
        d.residual = residual.data.data_ptr() if residual is not None else None
        d.residual_ld = layer.cout
        d.scale = d.shift = d.epi_label = None
        d.act = 0
        st, off, ld = out_ptr_ld
        d.out_raw, d.out_raw_ld = st.data_ptr() + 4 * off, ld
        d.out_act, d.out_act_ld = None, layer.cout
        d.tile_hint = 0
        d.head_weights = d.head_out = None
        d.head_cout = d.head_out_ld = 0
        d.head_label_out, d.head_label_classes = None, 0
        self.dy_ptr_ld = dy_ptr_ld
        # the full-resolution 1x1 heads (32 -> K / ver_dim): streaming kernels on the Keras kernel itself (csrc/head1x1.hip) unless CASAPOSE_HEAD_CONV=generic
        self.head_fast = (k == 1 and stride == 1 and pad == 0 and len(srcs) == 1 and layer.sources[0] == (32, 32) and layer.cout <= 32
                          and tap_label is None and row_scale is None and residual is None and srcs[0][1] % 4 == 0
                          and os.environ.get("CASAPOSE_HEAD_CONV", "stream") != "generic")
        self._out_ptr_ld = out_ptr_ld
        self.record_prefix = None   # (tensor, ld, floats): dense rows this head copies in front of its own columns (TrainPlan._whole_records)
        # fused normalisation around a Winograd layer (TrainPlan._fuse_winograd): `stats_to` = the normalisation op whose batch statistics this
        # op's output transform accumulates; `pre_norm[s]` = the normalisation op whose normalise + activate this op's input transform of
        # source s applies itself (its stored activation is then never written)
        self.stats_to: Optional["BnActOp"] = None
        self.pre_norm: Dict[int, "BnActOp"] = {}
        self.pre_bn: Optional["BnActOp"] = None   # fused normalisation (TrainPlan._fuse_heads): this head reads the RAW tensor and applies pre_bn's tables itself
        # data-gradient descriptors
        for s, ent in enumerate(layer.dgrad):
            if ent is None:
                continue
            t, ld = srcs[s]
            assert t.needs_grad and ld == t.c
            g = ent["desc"]
            g.batch = batch
            g.in_h, g.in_w = (in_h + 2 * pad - (eff - 1), in_w + 2 * pad - (eff - 1)) if stride == 2 else (self.out_h, self.out_w)
            if stride == 2:
                assert dilation == 1 and g.in_h == 2 * self.out_h and g.in_w == 2 * self.out_w, "stride-2 data gradient expects even geometry"
            g.out_h, g.out_w = in_h, in_w
            g.cout, g.kh, g.kw, g.stride, g.dilation, g.pad = ent["cout"], k, k, 1, dilation, dilation * (k - 1) - pad
            g.num_sources = 1
            g.src[0].channels = ent["cin"]
            g.src[0].mode = _lib.SRC_ZERO_INSERT_X2 if stride == 2 else _lib.SRC_DIRECT
            g.src[0].sel = g.src[0].pre_scale = g.src[0].pre_shift = None
            g.weights, g.weights_halo = ent["w"].data_ptr(), _ptr(ent["w_halo"])
            g.tap_label, g.row_scale = _ptr(tap_label), None
            g.residual_ld = ent["cout"]
            g.scale = g.shift = g.epi_label = None
            g.act = 0
            g.out_raw, g.out_raw_ld = t.grad.data_ptr(), t.c
            g.out_act, g.out_act_ld = None, t.c
            g.tile_hint = 0
            g.head_weights = g.head_out = None
            g.head_cout = g.head_out_ld = 0
            g.head_label_out, g.head_label_classes = None, 0

    # ---- 1x1 / stride-1 shortcuts as plain GEMMs on the bf16 matrix pipe -----------------------------------------------------------------
    def setup_gemm(self):
        """The stage shortcuts with stride 1 (resnet.py:214-220 at output stride 8: stage 1, 3 and 4) are GEMMs rows x cin x cout.  In the conv modes
        `split` / `bf16` the forward and the data gradient run on the bf16-pipe GEMM of the Winograd path (csrc/wino_gemm_split.hip: exact
        three-way splits, or hi + mid planes in the bf16 mode) and, where cin and cout are multiples of 128, the weight gradient on its
        transposed partner (csrc/wino_wgrad_split.hip) -- instead of the fp32-MFMA implicit-GEMM kernel (84 TFLOP/s on the 256 -> 512 shortcut).
        CASAPOSE_SC_GEMM=0 keeps the fp32 kernels."""
        L = self.layer
        self.gemm = None
        planes = self.layer.mode_planes
        if not planes or os.environ.get("CASAPOSE_SC_GEMM", "1") == "0":
            return
        rows = self.batch * self.out_h * self.out_w
        cin = L.sources[0][0]
        st_, off_, ld_ = self._out_ptr_ld
        if (L.k != 1 or self.stride != 1 or self.pad != 0 or len(self.srcs) != 1 or L.sources[0][1] != cin or self.srcs[0][1] != cin or cin % 32
                or L.cout % 32 or rows % 128 or self.residual is not None or self.row_scale is not None or self.tap_label is not None
                or self.head_fast or ld_ != L.cout or L.layout != 0 or self.out is None
                or (self.dy_ptr_ld is not None and (self.dy_ptr_ld[1] != 0 or self.dy_ptr_ld[2] != L.cout))):
            return
        lib = _lib.load()
        dev = L.master.device
        idx_t = (np.arange(cin, dtype=np.int32)[None, :] * L.cout + np.arange(L.cout, dtype=np.int32)[:, None]).reshape(-1)   # [co][ci] -> master [ci][co]
        g = dict(rows=rows, cin=cin, planes=3 if planes == 3 else 2, fwd_planes=_lib.PLANES_F16X2 if L.fwd_f16x2 else (3 if planes == 3 else 2), c_scale=1.0, cache={},
                 idx_t=torch.from_numpy(idx_t).to(dev),
                 Uf=torch.empty(L.cout * cin, dtype=torch.float32, device=dev),
                 Us_f=torch.empty(lib.cp_wino_split_weights_bytes(1, L.cout, cin), dtype=torch.uint8, device=dev),
                 Us_d=torch.empty(lib.cp_wino_split_weights_bytes(1, cin, L.cout), dtype=torch.uint8, device=dev) if L.dgrad[0] is not None else None,
                 wgrad=bool(lib.cp_wino_wgrad_split_applicable(1, rows, L.cout, cin)) and os.environ.get("CASAPOSE_WINO_WGRAD", "split") != "f32")
        if g["wgrad"]:
            g["dU"] = torch.empty(L.cout * cin, dtype=torch.float32, device=dev)
        self.gemm = g
        L.refresh_hooks.append(self._refresh_gemm)

    def _refresh_gemm(self, stream: int):
        from .engine import split_wino_weights

        lib = _lib.load()
        L, g = self.layer, self.gemm
        check(lib.cp_gather_f32(L.master.data_ptr(), g["idx_t"].data_ptr(), g["idx_t"].numel(), g["Uf"].data_ptr(), stream), "cp_gather_f32(%s)" % L.name)
        if g["fwd_planes"] == _lib.PLANES_F16X2:   # forward in the fp16 two-way split: U times a power of two, the GEMM undoes it
            scale = f16x2_scale(g["cache"], g["Uf"])
            g["c_scale"] = 1.0 / scale
            check(lib.cp_wino_split_weights_scaled_f32(g["Uf"].data_ptr(), 1, L.cout, g["cin"], _lib.PLANES_F16X2, scale, g["Us_f"].data_ptr(), stream),
                  "cp_wino_split_weights_scaled_f32(%s)" % L.name)
        else:
            split_wino_weights(g["Uf"], 1, L.cout, g["cin"], out=g["Us_f"], stream=stream)          # forward: U[co][ci] = W[ci][co]
        if g["Us_d"] is not None:
            split_wino_weights(L.master, 1, g["cin"], L.cout, out=g["Us_d"], stream=stream)     # data gradient: U[ci][co] = W[ci][co], the master itself

    # ---- Winograd F(4x4,3x3) for the deep layers (csrc/wino.hip): forward and data gradient ------------------------
    def setup_winograd(self):
        """Decide which of this op's convolutions (forward, per-source data gradient) take the Winograd path and return the
        scratch sizes (floats of V, floats of M) they need; bind_winograd() completes the descriptors."""
        from .engine import TRAIN_WINO_GEMM_SPLIT, WinoConv, wino_eligible

        L = self.layer
        self.wino_fwd = None
        self.wino_dgrad: Dict[int, dict] = {}
        nv = nm = 0
        if L.k != 3 or self.stride != 1 or self.pad != self.dil or self.tap_label is not None or L.layout != 0:
            return 0, 0
        _, tp = WinoConv.tiles(self.batch, self.in_h, self.in_w, self.dil)
        dev = L.master.device
        cin = sum(s[1] for s in L.sources)
        # The training plan keeps the K >= 256 threshold of the fp32 GEMM.  Measured in round 3 with the dilated 128 -> 256 layer
        # (stage3_unit1_conv1) on the Winograd path as in the inference plan: no gain on the step (510 vs 519 images/s: its weight gradient
        # falls back to the fp32 grouped GEMM), and the F(4x4,3x3) rounding of one more layer in the forward moved the config-13 gradient
        # comparison from 6e-5 to 3e-3 (the proxy-voting loss divides by |v|^2: gradients are very sensitive to the forward's last bits).
        # In the bf16 conv mode (3e-2 gates) the same layer does take the Winograd path: no bf16-pipe kernel covers a dilated 3x3 directly, so it
        # would otherwise be the one deep layer left on the fp32 MFMA (forward 0.60 ms at 99 TFLOP/s).
        # Round 6: the fp32-LEVEL default (forward in the fp16 two-way split, the config-13 comparison gated at 1e-2 like the fp32 MFMA's) takes it too:
        # forward 0.59 -> Winograd GEMM on fp16 pairs, weight gradient through the Winograd planes instead of the fp32 MFMA kernel.
        split_gemm = self.dil > 1 and (self.layer.mode_planes == 1 or (self.layer.fwd_f16x2 and os.environ.get("CASAPOSE_TRAIN_WINO_DILATED_128", "1") == "1"))

        if all(s[0] == s[1] for s in L.sources) and wino_eligible(3, 1, self.dil, self.pad, L.sources, L.cout, split_gemm=split_gemm):
            ktot = sum(s[0] for s in L.sources)
            # the forward's transformed input is kept per layer (not in the shared scratch): the weight gradient multiplies it again
            self.wino_fwd = dict(U=torch.zeros(36 * L.cout * ktot, dtype=torch.float32, device=dev), ktot=ktot, cout=L.cout, tp=tp, desc=ConvDesc(),
                                 V=torch.zeros(36 * tp * ktot, dtype=torch.float32, device=dev),  # padding tiles stay zero
                                 dU=torch.empty(36 * L.cout * ktot, dtype=torch.float32, device=dev), wdesc=ConvDesc())
            nv, nm = max(nv, 36 * tp * ktot), max(nm, 36 * tp * L.cout)
            if L.mode_planes == 3 and TRAIN_WINO_GEMM_SPLIT and train_bwd_f16x2():   # the weight-gradient GEMM in f16x2: transformed dY x 2^e
                self.wino_fwd["wg16"] = dict(e=None, mon=None, dead=False)
        c0 = 0
        for s, (ent, (cp, cr)) in enumerate(zip(L.dgrad, L.sources)):
            if ent is not None and L.cout % 32 == 0 and wino_eligible(3, 1, self.dil, self.dil, [(L.cout, L.cout)], cr, split_gemm=split_gemm):
                self.wino_dgrad[s] = dict(U=torch.zeros(36 * cr * L.cout, dtype=torch.float32, device=dev), ktot=L.cout, cout=cr, tp=tp, desc=ConvDesc(), c0=c0)
                if L.mode_planes == 3 and TRAIN_WINO_GEMM_SPLIT and train_bwd_f16x2():   # (e = None while the exact split runs and the slot measures)
                    self.wino_dgrad[s]["f16"] = dict(e=None, mon=None, dead=False)
                nv, nm = max(nv, 36 * tp * L.cout), max(nm, 36 * tp * cr)
            c0 += cr
        if self.wino_fwd is not None or self.wino_dgrad:
            L.refresh_hooks.append(self._refresh_winograd)
        self._cin = cin
        return nv, nm

    def _refresh_winograd(self, stream: int):
        lib = _lib.load()
        L = self.layer
        cin, cout = self._cin, L.cout
        m = L.master.data_ptr()
        if self.wino_fwd is not None:
            c0 = k0 = 0
            for cp, cr in L.sources:  # master is HWIO [3][3][cin][cout]
                check(lib.cp_wino_transform_weights_f32(m + 4 * c0 * cout, 3 * cin * cout, cin * cout, cout, 1, 0, cr, cout, self.wino_fwd["ktot"], k0,
                                                        self.wino_fwd["U"].data_ptr(), stream), "cp_wino_transform_weights_f32")
                c0 += cr
                k0 += cp
        for s, w in self.wino_dgrad.items():  # flipped taps, K index = forward output channel, output = forward input channel
            check(lib.cp_wino_transform_weights_f32(m + 4 * w["c0"] * cout, 3 * cin * cout, cin * cout, 1, cout, 1, cout, w["cout"], w["ktot"], 0,
                                                    w["U"].data_ptr(), stream), "cp_wino_transform_weights_f32")
        from .engine import TRAIN_WINO_GEMM_SPLIT, split_wino_weights

        if TRAIN_WINO_GEMM_SPLIT:  # GEMM on the bf16 matrix pipe (exact 3-way splits, fp32-equivalent): its weights are the pre-split planes of U
            for w in ([self.wino_fwd] if self.wino_fwd is not None else []) + list(self.wino_dgrad.values()):
                if w is self.wino_fwd and L.fwd_f16x2:   # the forward GEMM in the fp16 two-way split (the data gradients keep the exact bf16 split)
                    if w.get("Us") is None:
                        w["Us"] = torch.empty(lib.cp_wino_split_weights_bytes(36, w["cout"], w["ktot"]), dtype=torch.uint8, device=w["U"].device)
                        w["cache"] = {}
                    scale = f16x2_scale(w["cache"], w["U"])
                    w["c_scale"] = 1.0 / scale
                    check(lib.cp_wino_split_weights_scaled_f32(w["U"].data_ptr(), 36, w["cout"], w["ktot"], _lib.PLANES_F16X2, scale, w["Us"].data_ptr(), stream),
                          "cp_wino_split_weights_scaled_f32(%s)" % L.name)
                elif w.get("f16") is not None and w["f16"]["e"] is not None:   # a data gradient in the fp16 two-way split (train_bwd_f16x2)
                    if w.get("Us16") is None:
                        w["Us16"] = torch.empty(lib.cp_wino_split_weights_bytes(36, w["cout"], w["ktot"]), dtype=torch.uint8, device=w["U"].device)
                        w["cache"] = {}
                    scale = f16x2_scale(w["cache"], w["U"])
                    w["c_scale"] = 2.0 ** (-w["f16"]["e"]) / scale
                    check(lib.cp_wino_split_weights_scaled_f32(w["U"].data_ptr(), 36, w["cout"], w["ktot"], _lib.PLANES_F16X2, scale, w["Us16"].data_ptr(), stream),
                          "cp_wino_split_weights_scaled_f32(dgrad %s)" % L.name)
                    w["Us"] = w["Us16"]
                else:
                    w.pop("c_scale", None)
                    w["Us"] = w["Us3"] = split_wino_weights(w["U"], 36, w["cout"], w["ktot"], out=w.get("Us3"), stream=stream)

    def wgrad_f16x2(self) -> bool:
        """the direct weight gradient on fp16 pairs: dY inside the band (the switch its data gradient uses) and X too (the forward still f16x2)"""
        b = getattr(self, "bw16", None)
        return b is not None and b["on"] and not b["dead"] and self.layer.fwd_f16x2

    def direct_dgrad_split(self) -> bool:
        """True when this op has a data gradient that runs on conv_hsplit as an exact three-way split and may move to the fp16 pair
        (train_bwd_f16x2): 3x3 / stride 1 / no dilation, not a Winograd / deep-bf16 / 1x1-GEMM route"""
        L = self.layer
        if not (train_bwd_f16x2() and L.mode_planes == 3 and L.k == 3 and self.stride == 1 and self.dil == 1) or getattr(self, "gemm", None) is not None:
            return False
        for s, ent in enumerate(L.dgrad):
            if ent is None or ent["split"] is None or ent.get("deep") or s in getattr(self, "wino_dgrad", {}):
                continue
            if ent["split"]["np"] in (3, _lib.PLANES_F16X2):
                return True
        return False

    def set_direct_dgrad_f16x2(self, on: bool, stream: int):
        """the direct data gradients of this op on the fp16 two-way split (weights x 2^k re-packed as two fp16 planes) or back on the exact split"""
        L = self.layer
        self.bw16["on"] = on
        for s, ent in enumerate(L.dgrad):
            if ent is None or ent["split"] is None or ent.get("deep") or s in getattr(self, "wino_dgrad", {}):
                continue
            sp = ent["split"]
            if sp["np"] in (3, _lib.PLANES_F16X2):
                sp["np"] = _lib.PLANES_F16X2 if on else 3
        L._refresh_split(stream)

    def set_dgrad_exponent(self, w: dict, e: Optional[int], stream: int):
        """the power of two the transformed dY of this Winograd data gradient is multiplied by (None: back to the exact split); re-packs the weights"""
        f = w["f16"]
        f["e"] = e
        if e is not None:
            if w.get("ps") is None:
                dev = w["U"].device
                w["ps"] = torch.empty(w["ktot"], dtype=torch.float32, device=dev)
            w["ps"].fill_(2.0 ** e)
        self._refresh_winograd(stream)

    def bind_winograd(self, V: torch.Tensor, M: torch.Tensor, M2: Optional[torch.Tensor] = None):
        self._wV, self._wM, self._wM2 = V, M, (M2 if M2 is not None else M)
        for w in ([self.wino_fwd] if self.wino_fwd is not None else []) + list(self.wino_dgrad.values()):
            d = w["desc"]
            d.batch, d.in_h, d.in_w, d.out_h, d.out_w = 36, 1, w["tp"], 1, w["tp"]
            d.cout, d.kh, d.kw, d.stride, d.dilation, d.pad = w["cout"], 1, 1, 1, 1, 0
            d.num_sources = 1
            d.src[0].data, d.src[0].channels, d.src[0].ld, d.src[0].mode = V.data_ptr(), w["ktot"], w["ktot"], _lib.SRC_DIRECT
            d.weights = w["U"].data_ptr()
            d.out_raw, d.out_raw_ld, d.out_act_ld, d.residual_ld = M.data_ptr(), w["cout"], w["cout"], w["cout"]
            d.group_rows, d.group_weight_stride = w["tp"], w["cout"] * w["ktot"]
        if self.wino_fwd is not None:  # grouped 1x1 weight-gradient problem over the 36 planes: dU[p] = dM[p]^T V[p]
            w = self.wino_fwd
            g = w["wdesc"]
            g.batch, g.in_h, g.in_w, g.out_h, g.out_w = 1, 1, 36 * w["tp"], 1, 36 * w["tp"]
            g.cout, g.kh, g.kw, g.stride, g.dilation, g.pad = w["cout"], 1, 1, 1, 1, 0
            g.num_sources = 1
            g.src[0].data, g.src[0].channels, g.src[0].ld, g.src[0].mode = w["V"].data_ptr(), w["ktot"], w["ktot"], _lib.SRC_DIRECT
            g.group_rows = w["tp"]

    def _wino_run(self, w, srcs, residual_ptr, out_ptr, stream, pre=None, stats=None, mon=None):
        """srcs: list of (ptr, ld, channels); writes out_ptr[pix][w.cout] = conv (+ residual).  pre: {source index: (scale ptr, shift ptr, act)}
        applied by the input transform; stats: fp64 [2][cout] table the output transform fills with the output's batch statistics; mon: f16x2 range
        monitor slot armed around the input transforms (they report max |V| of what the f16x2 GEMM will convert; the GEMM itself stays un-armed:
        the padding rows of V hold stale data)."""
        lib = _lib.load()
        off = 0
        V = w["V"] if "V" in w else self._wV
        if mon:
            lib.cp_f16x2_monitor_set(mon)
        try:
            for i, (ptr, ld, ch) in enumerate(srcs):
                ps, pb, pa = (pre or {}).get(i, (None, None, 0))
                check(lib.cp_wino_input_transform_pre_f32(ptr, ld, ch, self.batch, self.in_h, self.in_w, self.dil, V.data_ptr(), w["ktot"], off, ps, pb, pa, stream),
                      "cp_wino_input_transform_pre_f32(%s)" % self.layer.name)
                off += ch
        finally:
            if mon:
                lib.cp_f16x2_monitor_set(None)
        if w.get("Us") is not None:
            # CASAPOSE_CONV_MODE=bf16: hi + mid planes only (three products, not fp32-equivalent: that mode's gates are 3e-2); else the exact split
            f16x2 = "c_scale" in w
            check(lib.cp_wino_gemm_split_scaled_f32(V.data_ptr(), w["Us"].data_ptr(), self._wM.data_ptr(), 36 * w["tp"], w["tp"], w["ktot"], w["cout"],
                                                    _lib.PLANES_F16X2 if f16x2 else (2 if self.layer.mode_planes == 1 else 3), w["c_scale"] if f16x2 else 1.0, stream),
                  "cp_wino_gemm_split_scaled_f32(%s)" % self.layer.name)
        else:
            check(lib.cp_wino_gemm_f32(V.data_ptr(), w["U"].data_ptr(), self._wM.data_ptr(), 36 * w["tp"], w["tp"], w["ktot"], w["cout"], stream),
                  "cp_wino_gemm_f32(%s)" % self.layer.name)
        check(lib.cp_wino_output_transform_stats_f32(self._wM.data_ptr(), w["cout"], self.batch, self.in_h, self.in_w, self.dil, residual_ptr, w["cout"], None, None,
                                                     None, 0, out_ptr, w["cout"], None, w["cout"], stats, stream),
              "cp_wino_output_transform_stats_f32(%s)" % self.layer.name)

    def forward(self, stream: int):
        """The forward launch(es) of this layer; an f16x2 forward runs armed when the plan has given the op a monitor slot (mon_ptr)."""
        mon = getattr(self, "mon_ptr", None) if self.layer.fwd_f16x2 else None
        if getattr(self, "wino_fwd", None) is not None:
            return self._forward(stream, mon)
        if not mon:
            return self._forward(stream, None)
        lib = _lib.load()
        lib.cp_f16x2_monitor_set(mon)
        try:
            self._forward(stream, None)
        finally:
            lib.cp_f16x2_monitor_set(None)

    def demote_forward_to_exact_split(self, stream: int):
        """this op's forward on the exact three-way bf16 split from now on (its f16x2 operands left the fp16 range condition): re-packs the forward
        weights of whichever route the op takes -- the direct / stem kernel's planes, the 1x1 GEMM's, the Winograd GEMM's"""
        L = self.layer
        if not L.fwd_f16x2:
            return
        L.fwd_f16x2 = False
        sp = L.split
        if sp is not None and sp["np"] == _lib.PLANES_F16X2:
            sp["np"], sp["descale"] = 3, 1.0
            sp["planes"] = torch.empty(sp["f32"].numel() // 512 * 3 * 1024, dtype=torch.uint8, device=sp["f32"].device)
            L._refresh_split(stream)
        g = getattr(self, "gemm", None)
        if g is not None and g["fwd_planes"] == _lib.PLANES_F16X2:
            g["fwd_planes"], g["c_scale"] = g["planes"], 1.0
            self._refresh_gemm(stream)
        w = getattr(self, "wino_fwd", None)
        if w is not None and "c_scale" in w:
            del w["c_scale"]
            self._refresh_winograd(stream)

    def _forward(self, stream: int, mon):
        if getattr(self, "gemm", None) is not None:
            g, d = self.gemm, self.layer.desc
            check(_lib.load().cp_wino_gemm_split_scaled_f32(self.srcs[0][0].data.data_ptr(), g["Us_f"].data_ptr(), d.out_raw, g["rows"], g["rows"], g["cin"],
                                                            self.layer.cout, g["fwd_planes"], g["c_scale"], stream), "cp_wino_gemm_split_scaled_f32(%s)" % self.layer.name)
            return
        if getattr(self, "wino_fwd", None) is not None:
            d = self.layer.desc
            srcs, pre = [], {}
            for i, ((t, ld), c) in enumerate(zip(self.srcs, self.layer.sources)):
                bn = self.pre_norm.get(i)
                if bn is not None:   # read the RAW tensor; normalise + activate while loading
                    srcs.append((bn.x.data.data_ptr(), bn.x.c, c[0]))
                    pre[i] = (bn.scale.data_ptr(), bn.shift.data_ptr(), bn.act)
                else:
                    srcs.append((t.data.data_ptr(), ld, c[0]))
            self._wino_run(self.wino_fwd, srcs, self.residual.data.data_ptr() if self.residual is not None else None, d.out_raw, stream, pre=pre,
                           stats=self.stats_to.sums.data_ptr() if self.stats_to is not None else None, mon=mon)
            return
        lib = _lib.load()
        if self.head_fast and self.pre_bn is not None and self.record_prefix is not None:
            bn = self.pre_bn
            st_, off, old_ = self._out_ptr_ld
            pre, pre_ld, pre_n = self.record_prefix
            assert off == pre_n
            check(lib.cp_head1x1_fwd_affine_record_f32(bn.x.data.data_ptr(), bn.x.c, self.batch * self.out_h * self.out_w, bn.scale.data_ptr(), bn.shift.data_ptr(),
                                                       _ptr(bn.labels), bn.classes, bn.act, self.layer.master.data_ptr(), self.layer.cout,
                                                       pre.data_ptr(), pre_ld, pre_n, st_.data_ptr(), old_, stream),
                  "cp_head1x1_fwd_affine_record_f32(%s)" % self.layer.name)
            return
        if self.head_fast and self.pre_bn is not None:
            bn = self.pre_bn
            st_, off, old_ = self._out_ptr_ld
            check(lib.cp_head1x1_fwd_affine_f32(bn.x.data.data_ptr(), bn.x.c, self.batch * self.out_h * self.out_w, bn.scale.data_ptr(), bn.shift.data_ptr(),
                                                _ptr(bn.labels), bn.classes, bn.act, self.layer.master.data_ptr(), self.layer.cout,
                                                st_.data_ptr() + 4 * off, old_, stream), "cp_head1x1_fwd_affine_f32(%s)" % self.layer.name)
            return
        if self.head_fast:
            t, ld = self.srcs[0]
            st_, off, old_ = self._out_ptr_ld
            check(lib.cp_head1x1_fwd_f32(t.data.data_ptr(), ld, self.batch * self.out_h * self.out_w, self.layer.master.data_ptr(), self.layer.cout,
                                         st_.data_ptr() + 4 * off, old_, stream), "cp_head1x1_fwd_f32(%s)" % self.layer.name)
            return
        sp = self.layer.split
        if sp is not None and sp.get("stem"):
            if getattr(self, "_stem_fwd", None) is None:   # stride 2 / pad 3 / one 4-channel source: the range of the stem kernels
                self._stem_fwd = lib.cp_conv_selected_tile(C.byref(self.layer.desc)) == _lib.TILE_STEM
            if self._stem_fwd:
                check(lib.cp_conv2d_fwd_stem_split_scaled(C.byref(self.layer.desc), sp["planes"].data_ptr(), sp["np"], sp["descale"], stream),
                      "cp_conv2d_fwd_stem_split(%s)" % self.layer.name)
                return
        elif sp is not None:
            if getattr(self, "_split_fwd", None) is None:
                self._split_fwd = bool(lib.cp_conv_split_applicable(C.byref(self.layer.desc)))
            if self._split_fwd:
                check(lib.cp_conv2d_fwd_split_scaled(C.byref(self.layer.desc), sp["planes"].data_ptr(), None, sp["np"], sp["descale"], 1.0, stream),
                      "cp_conv2d_fwd_split(%s)" % self.layer.name)
                return
        check(lib.cp_conv2d_fwd_f32(C.byref(self.layer.desc), stream), "cp_conv2d_fwd_f32(%s)" % self.layer.name)

    def executed_flops(self) -> Dict[str, float]:
        """FLOPs this op's launches EXECUTE per step, by matrix pipe: {"f32": fp32-MFMA FLOPs, "bf16": bf16-MFMA FLOPs} -- a Winograd layer counts
        its grouped GEMMs (36 planes x padded tiles), an exact three-way split counts six bf16 products per fp32 product; forward + weight
        gradient + every data gradient.  (bench.py --mode train prices the step against the two pipes' peaks with it.)"""
        from .engine import TRAIN_WINO_GEMM_SPLIT

        L = self.layer
        lib = _lib.load()
        out = {"f32": 0.0, "bf16": 0.0}
        m_out = float(self.batch * self.out_h * self.out_w)
        cin = sum(s[1] for s in L.sources)
        direct = 2.0 * m_out * L.k * L.k * cin * L.cout
        wino_pipe, wino_mult = ("bf16", 3.0 if self.layer.mode_planes == 1 else 6.0) if TRAIN_WINO_GEMM_SPLIT else ("f32", 1.0)

        def split_pipe(sp):
            return ("bf16", {3: 6.0, _lib.PLANES_F16X2: 3.0}.get(sp["np"], 1.0))   # products per fp32 product: exact bf16 split, fp16 two-way split, bf16

        if getattr(self, "gemm", None) is not None:
            gm = self.gemm
            mult = 6.0 if gm["planes"] == 3 else 3.0
            out["bf16"] += (3.0 if gm["fwd_planes"] == _lib.PLANES_F16X2 else mult) * direct     # forward
            if gm["Us_d"] is not None:
                out["bf16"] += mult * direct                                                      # data gradient
            if gm["wgrad"]:
                out["bf16"] += (1.0 if self.layer.mode_planes == 1 else 6.0) * direct
            else:
                out["f32"] += direct
            return out
        if getattr(self, "wino_fwd", None) is not None:
            w = self.wino_fwd
            g = 2.0 * 36 * w["tp"] * w["ktot"] * w["cout"]
            out[wino_pipe] += (3.0 if (L.fwd_f16x2 and TRAIN_WINO_GEMM_SPLIT) else wino_mult) * g      # forward GEMM
            if self.wino_wgrad_split():          # weight gradient: grouped GEMM over the 36 planes, exact splits / f16x2 on the 2-byte pipe or fp32 MFMA
                wg16 = w.get("wg16") is not None and w["wg16"]["e"] is not None and L.fwd_f16x2
                out["bf16"] += (1.0 if self.layer.mode_planes == 1 else (3.0 if wg16 else 6.0)) * g
            else:
                out["f32"] += g
        else:
            if L.split is not None and (lib.cp_conv_split_applicable(C.byref(L.desc)) or (L.split.get("stem") and getattr(self, "_stem_fwd", True))):
                pipe, mult = split_pipe(L.split)
                out[pipe] += mult * direct
            else:
                out["f32"] += direct
            wp = self.wgrad_planes()             # weight gradient: conv_wgrad_split.hip (32-multiple sources; the image source stays fp32) or fp32 MFMA
            if wp:
                big = sum(s[1] for s in L.sources if s[0] != 4)
                out["bf16"] += ((3.0 if self.wgrad_f16x2() else 6.0) if wp == 3 else 1.0) * direct * big / cin
                out["f32"] += direct * (cin - big) / cin
            else:
                out["f32"] += direct
        c0 = 0
        for s, (ent, (cp, cr)) in enumerate(zip(L.dgrad, L.sources)):
            if ent is None:
                continue
            if ent.get("deep") and ent["split"] is not None:
                out["bf16"] += 2.0 * float(self.batch * self.in_h * self.in_w) * L.k * L.k * ent["cin"] * cr
            elif s in getattr(self, "wino_dgrad", {}):
                w = self.wino_dgrad[s]
                f16 = w.get("f16") is not None and w["f16"]["e"] is not None
                out[wino_pipe] += (3.0 if f16 else wino_mult) * 2.0 * 36 * w["tp"] * w["ktot"] * w["cout"]
            else:
                m_in = float(self.batch * self.in_h * self.in_w)
                d = 2.0 * m_in * L.k * L.k * ent["cin"] * cr
                if ent["split"] is not None and self.stride == 1 and self.dil == 1 and L.k == 3:
                    pipe, mult = split_pipe(ent["split"])
                    out[pipe] += mult * d
                else:
                    out["f32"] += d
        return out

    def wino_wgrad_split(self) -> bool:
        """True when this Winograd layer's weight-gradient GEMM runs on the bf16 matrix pipe: the default wherever the Winograd GEMMs do
        (CASAPOSE_WINO_GEMM != f32) and the shape fits (cout, cin multiples of 128); CASAPOSE_WINO_WGRAD=f32 keeps the fp32 grouped GEMM."""
        from .engine import TRAIN_WINO_GEMM_SPLIT

        w = getattr(self, "wino_fwd", None)
        if w is None or not TRAIN_WINO_GEMM_SPLIT or os.environ.get("CASAPOSE_WINO_WGRAD", "split") == "f32":
            return False
        return bool(_lib.load().cp_wino_wgrad_split_applicable(36, w["tp"], self.layer.cout, w["ktot"]))

    def wgrad_planes(self) -> int:
        """3 / 1 when this op's weight gradient runs on the bf16 matrix pipe (CASAPOSE_CONV_MODE split / bf16 and a descriptor that
        cp_conv2d_wgrad_split covers: 3x3 / stride 1 / pad 1, 32-multiple sources + optional image, cout % 32 == 0), else 0 = fp32 MFMA."""
        planes = self.layer.mode_planes
        if not planes or getattr(self, "wino_fwd", None) is not None:
            return 0
        return planes if _lib.load().cp_conv_wgrad_split_applicable(C.byref(self.layer.desc)) else 0

    def _dy(self):
        if self.dy_ptr_ld is not None:
            st, off, ld = self.dy_ptr_ld
            return st.data_ptr() + 4 * off, ld
        assert self.out.has_grad, "gradient of %s not produced" % self.layer.name
        return self.out.grad.data_ptr(), self.out.c

    def backward(self, stream: int, wgrad_stream: Optional[int] = None):
        """weight gradient, then the data gradient of every source.  wgrad_stream (TrainPlan.backward with a side stream): the weight-gradient
        launches go there -- they are off the critical path of the backward (nothing but Adam reads them) and MFMA-bound, so they can run under
        the HBM-bound normalisation / resampling passes of the layers before; the caller orders the two streams with events."""
        self.backward_wgrad(stream if wgrad_stream is None else wgrad_stream, side=wgrad_stream is not None)
        self.backward_dgrad(stream)

    def backward_wgrad(self, stream: int, side: bool = False):
        lib = _lib.load()
        L = self.layer
        dy, dy_ld = self._dy()
        d = L.desc  # one op per layer: filled by this op's constructor
        if self.head_fast and self.pre_bn is not None:   # the activated input is recomputed from the raw tensor
            bn = self.pre_bn
            px = self.batch * self.out_h * self.out_w
            check(lib.cp_head1x1_wgrad_affine_f32(bn.x.data.data_ptr(), bn.x.c, bn.scale.data_ptr(), bn.shift.data_ptr(), _ptr(bn.labels), bn.classes, bn.act,
                                                  dy, dy_ld, px, L.cout, L.master_grad.data_ptr(), 1 if self.accumulate_master else 0, stream),
                  "cp_head1x1_wgrad_affine_f32(%s)" % L.name)
            return
        if self.head_fast:
            t, ld = self.srcs[0]
            px = self.batch * self.out_h * self.out_w
            check(lib.cp_head1x1_wgrad_f32(t.data.data_ptr(), ld, dy, dy_ld, px, L.cout, L.master_grad.data_ptr(), 1 if self.accumulate_master else 0, stream),
                  "cp_head1x1_wgrad_f32(%s)" % L.name)
            return
        if getattr(self, "gemm", None) is not None and self.gemm["wgrad"]:
            # dU[co][ci] = sum_rows dY[row][co] A[row][ci] on the bf16 pipe, then through the transpose map into the master gradient [ci][co]
            g = self.gemm
            check(lib.cp_wino_wgrad_split_f32(dy, self.srcs[0][0].data.data_ptr(), g["dU"].data_ptr(), 1, g["rows"], L.cout, g["cin"],
                                              1 if self.layer.mode_planes == 1 else 3, stream), "cp_wino_wgrad_split_f32(%s)" % L.name)
            check(lib.cp_scatter_f32(g["dU"].data_ptr(), g["idx_t"].data_ptr(), g["idx_t"].numel(), L.master_grad.data_ptr(), 1 if self.accumulate_master else 0,
                                     stream), "cp_scatter_f32(%s)" % L.name)
            return
        if getattr(self, "wino_fwd", None) is not None:
            # weight gradient through the Winograd planes: a quarter of the MFMA work of the direct kernel (V kept from the forward)
            w = self.wino_fwd
            cin, cout = self._cin, L.cout
            wM = self._wM2 if side else self._wM   # the side stream transforms dY into a scratch of its own (the data gradients use _wM)
            f = w.get("wg16") if self.wino_wgrad_split() else None
            arm = f is not None and not f["dead"] and f["mon"]
            if arm:
                lib.cp_f16x2_monitor_set(f["mon"])   # the transform reports max |dM|
            try:
                check(lib.cp_wino_dy_transform_f32(dy, dy_ld, cout, self.batch, self.in_h, self.in_w, self.dil, wM.data_ptr(), stream),
                      "cp_wino_dy_transform_f32(%s)" % L.name)
            finally:
                if arm:
                    lib.cp_f16x2_monitor_set(None)
            if self.wino_wgrad_split():   # the grouped GEMM dU[p] = dM[p]^T V[p] on the bf16 matrix pipe (exact splits; csrc/wino_wgrad_split.hip)
                # exact splits (fp32-equivalent) by default; CASAPOSE_CONV_MODE=bf16 rounds the operands of this GEMM to bf16 like the other weight gradients
                if f is not None and f["e"] is not None and L.fwd_f16x2:
                    # fp16 two-way split (train_bwd_f16x2): dM x 2^e from its monitor slot; V as it is -- the forward's monitor keeps it in the band
                    # (a forward that left the band is demoted: L.fwd_f16x2 turns False and this GEMM returns to the exact split with it)
                    check(lib.cp_wino_wgrad_split_scaled_f32(wM.data_ptr(), w["V"].data_ptr(), w["dU"].data_ptr(), 36, w["tp"], cout, w["ktot"],
                                                             _lib.PLANES_F16X2, 2.0 ** f["e"], 1.0, stream), "cp_wino_wgrad_split_scaled_f32(%s)" % L.name)
                else:
                    check(lib.cp_wino_wgrad_split_f32(wM.data_ptr(), w["V"].data_ptr(), w["dU"].data_ptr(), 36, w["tp"], cout, w["ktot"],
                                                      1 if self.layer.mode_planes == 1 else 3, stream),
                          "cp_wino_wgrad_split_f32(%s)" % L.name)
            else:
                check(lib.cp_conv2d_wgrad_f32(C.byref(w["wdesc"]), wM.data_ptr(), cout, w["dU"].data_ptr(), 0, stream), "cp_conv2d_wgrad_f32(wino %s)" % L.name)
            c0 = k0 = 0
            for cp_, cr in L.sources:
                check(lib.cp_wino_weight_grad_f32(w["dU"].data_ptr(), cr, cout, w["ktot"], k0, 3 * cin * cout, cin * cout, cout, 1,
                                                  L.master_grad.data_ptr() + 4 * c0 * cout, 1 if self.accumulate_master else 0, stream),
                      "cp_wino_weight_grad_f32(%s)" % L.name)
                c0 += cr
                k0 += cp_
        else:
            planes = self.wgrad_planes()
            if planes == 3 and self.wgrad_f16x2():
                planes = _lib.PLANES_F16X2   # both operands inside fp16's band: X watched by the forward's monitor, dY carrying the loss factor
            if planes:   # bf16 matrix pipe (csrc/conv_wgrad_split.hip): same packed result
                check(lib.cp_conv2d_wgrad_split(C.byref(d), dy, dy_ld, L.dwp.data_ptr(), 0, planes, stream), "cp_conv2d_wgrad_split(%s)" % L.name)
            else:
                check(lib.cp_conv2d_wgrad_f32(C.byref(d), dy, dy_ld, L.dwp.data_ptr(), 0, stream), "cp_conv2d_wgrad_f32(%s)" % L.name)
            check(lib.cp_scatter_f32(L.dwp.data_ptr(), L.idx_fwd.data_ptr(), L.idx_fwd.numel(), L.master_grad.data_ptr(), 1 if self.accumulate_master else 0,
                                     stream), "cp_scatter_f32")

    def backward_dgrad(self, stream: int):
        lib = _lib.load()
        L = self.layer
        dy, dy_ld = self._dy()
        if self.head_fast and self.pre_bn is not None:
            return   # pre_bn's backward recomputes this head's data gradient inside its two passes (cp_head1x1_bn_bwd_*)
        if self.head_fast:
            t, ld = self.srcs[0]
            px = self.batch * self.out_h * self.out_w
            readable = (self.dy_ptr_ld[2] - self.dy_ptr_ld[1] % self.dy_ptr_ld[2]) if self.dy_ptr_ld is not None else self.out.c
            if t.needs_grad:
                check(lib.cp_head1x1_dgrad_f32(dy, dy_ld, min(32, readable), px, L.master.data_ptr(), L.cout, t.grad.data_ptr(), t.c, 1 if t.has_grad else 0, stream),
                      "cp_head1x1_dgrad_f32(%s)" % L.name)
                t.has_grad = True
            return
        for s, ent in enumerate(L.dgrad):
            if ent is None:
                continue
            t, _ = self.srcs[s]
            if getattr(self, "gemm", None) is not None and self.gemm["Us_d"] is not None and not t.has_grad:
                # dA[row][ci] = sum_co dY[row][co] W[ci][co]; the GEMM writes (no accumulation): the plan runs this op's backward BEFORE the
                # other consumers of its input (TrainPlan puts the shortcut after conv1 in the tape), which then accumulate into it
                g = self.gemm
                check(lib.cp_wino_gemm_split_planes_f32(dy, g["Us_d"].data_ptr(), t.grad.data_ptr(), g["rows"], g["rows"], L.cout, g["cin"], g["planes"], stream),
                      "cp_wino_gemm_split_planes_f32(dgrad %s)" % L.name)
                t.has_grad = True
                continue
            if ent.get("deep") and ent["split"] is not None:
                g = ent["desc"]
                g.src[0].data, g.src[0].ld = dy, dy_ld
                g.residual = t.grad.data_ptr() if t.has_grad else None
                if lib.cp_conv_bf16_deep_applicable(C.byref(g)):
                    check(lib.cp_conv2d_fwd_bf16_deep(C.byref(g), ent["split"]["planes"].data_ptr(), stream), "dgrad bf16 deep(%s)" % L.name)
                    t.has_grad = True
                    continue
            if s in getattr(self, "wino_dgrad", {}):  # stride 1, so the data gradient lives on the forward's input grid
                w = self.wino_dgrad[s]
                f = w.get("f16")
                pre = {0: (w["ps"].data_ptr(), None, _lib.ACT_NONE)} if f is not None and f["e"] is not None else None   # (a factor only: no shift table)
                self._wino_run(w, [(dy, dy_ld, L.cout)], t.grad.data_ptr() if t.has_grad else None, t.grad.data_ptr(), stream, pre=pre,
                               mon=f["mon"] if f is not None and not f["dead"] else None)
                t.has_grad = True
                continue
            g = ent["desc"]
            g.src[0].data, g.src[0].ld = dy, dy_ld
            g.residual = t.grad.data_ptr() if t.has_grad else None
            sp = ent["split"]
            if sp is not None and lib.cp_conv_split_applicable(C.byref(g)):
                check(lib.cp_conv2d_fwd_split_scaled(C.byref(g), sp["planes"].data_ptr(), None, sp["np"], sp["descale"], 1.0, stream), "dgrad split(%s)" % L.name)
            else:
                check(lib.cp_conv2d_fwd_f32(C.byref(g), stream), "dgrad(%s)" % L.name)
            t.has_grad = True
        if self.residual is not None:
            add_grad(self.residual, dy, self.out.pixels * self.out.c, stream)


def add_grad(t: TT, src_ptr: int, n: int, stream: int):
    lib = _lib.load()
    if t.has_grad:
        check(lib.cp_axpby_f32(t.grad.data_ptr(), 1.0, src_ptr, 1.0, n, t.grad.data_ptr(), stream), "cp_axpby_f32")
    else:
        check(lib.cp_axpby_f32(src_ptr, 1.0, None, 0.0, n, t.grad.data_ptr(), stream), "cp_axpby_f32")
        t.has_grad = True


class BnActOp:
    """y = act(gamma[l]*(x-mean)*rstd + beta[l]) with batch statistics (optionally all-reduced across replicas)."""

    def __init__(self, plan: "TrainPlan", name: str, x: TT, y: TT, act: int, gamma: Optional[str], beta: Optional[str],
                 labels: Optional[torch.Tensor] = None, classes: int = 1, row_scale: Optional[torch.Tensor] = None,
                 pad_one: bool = False):
        self.plan, self.name, self.x, self.y, self.act = plan, name, x, y, act
        self.gamma_key, self.beta_key, self.labels, self.classes = gamma, beta, labels, classes
        self.row_scale = row_scale  # partial convolution: x = row_scale * conv, so d conv = row_scale * dx
        self.pad_one = pad_one      # bn_data: channel 3 is padding; its output is the constant 1 (see TrainPlan)
        C_ = x.c
        dev = x.data.device
        f64 = dict(dtype=torch.float64, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.sums = torch.zeros(2 * C_, **f64)
        self.mean = torch.zeros(C_, **f32)
        self.rstd = torch.zeros(C_, **f32)
        self.scale = torch.zeros(classes, C_, **f32)
        self.shift = torch.zeros(classes, C_, **f32)
        self.gamma_full = torch.ones(classes, C_, **f32)
        self.beta_full = torch.zeros(classes, C_, **f32)
        self._redchan = torch.zeros(classes * C_ * 2 + C_ * 2, **f64)   # adjacent: the backward reduce zeroes both with one fill
        self.red = self._redchan[:classes * C_ * 2]
        self.chan = self._redchan[classes * C_ * 2:]
        st = plan.store
        self.real_c = st.offsets[beta][1][-1] if beta else C_
        self.gamma_p = st.view(gamma) if gamma else None
        self.beta_p = st.view(beta) if beta else None
        self.dgamma_p = st.grad_view(gamma) if gamma else None
        self.dbeta_p = st.grad_view(beta) if beta else None
        self.mm = st.state.get(name + ".moving_mean")
        self.mv = st.state.get(name + ".moving_variance")
        # fused normalisation (TrainPlan._fuse_heads): `head` = the 1x1 head ConvOp that consumes y alone; y and its gradient are then never
        # written -- the head applies scale / shift itself and this op's backward recomputes the head's data gradient (csrc/head1x1.hip)
        self.head: Optional[ConvOp] = None
        # ... around Winograd layers (TrainPlan._fuse_winograd): `stats_from` = the convolution whose output transform has already filled
        # self.sums; `consumer` = the convolution whose input transform applies scale / shift / activation itself (y is never written)
        self.stats_from: Optional[ConvOp] = None
        self.consumer: Optional[ConvOp] = None

    def forward(self, stream: int):
        lib = _lib.load()
        x, p = self.x, self.plan
        C_ = x.c
        if self.stats_from is None:
            check(lib.cp_bn_stats_f32(x.data.data_ptr(), x.pixels, C_, C_, self.sums.data_ptr(), stream), "cp_bn_stats_f32(%s)" % self.name)
        n = p.all_reduce_stats(self.sums, x.pixels)
        upd = p.update_moving and self.mm is not None
        check(lib.cp_bn_finalize_f32(self.sums.data_ptr(), float(n), C_, self.real_c, self.classes, _ptr(self.gamma_p), _ptr(self.beta_p), BN_EPS,
                                     1 if self.pad_one else 0, BN_MOMENTUM, _ptr(self.mm) if upd else None, _ptr(self.mv) if upd else None,
                                     self.mean.data_ptr(), self.rstd.data_ptr(), self.gamma_full.data_ptr(), self.beta_full.data_ptr(),
                                     self.scale.data_ptr(), self.shift.data_ptr(), stream), "cp_bn_finalize_f32(%s)" % self.name)
        if self.head is None and self.consumer is None:
            self.materialize(stream)

    def materialize(self, stream: int):
        """y = act(x * scale[l] + shift[l]) as a stored tensor (the plain path; fused layers call it only for activation_pattern())."""
        x = self.x
        check(_lib.load().cp_affine_act_f32(x.data.data_ptr(), x.pixels, x.c, x.c, self.scale.data_ptr(), self.shift.data_ptr(), _ptr(self.labels), self.act,
                                            self.y.data.data_ptr(), x.c, stream), "cp_affine_act_f32(%s)" % self.name)

    def _backward_fused_head(self, stream: int):
        lib = _lib.load()
        x, p, hd = self.x, self.plan, self.head
        st_, off, ld = hd.dy_ptr_ld
        dout, readable = st_.data_ptr() + 4 * off, ld - off % ld
        args = (x.data.data_ptr(), x.c, dout, ld, min(32, readable), x.pixels, hd.layer.master.data_ptr(), hd.layer.cout, self.mean.data_ptr(),
                self.rstd.data_ptr(), self.gamma_full.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(), _ptr(self.labels), self.classes, self.act)
        check(lib.cp_head1x1_bn_bwd_reduce_f32(*args, self.red.data_ptr(), self.chan.data_ptr(), stream), "cp_head1x1_bn_bwd_reduce_f32(%s)" % self.name)
        n = p.all_reduce_stats(self.chan, x.pixels)
        assert x.needs_grad and not x.has_grad
        check(lib.cp_head1x1_bn_bwd_apply_f32(*args, self.chan.data_ptr(), float(n), _ptr(self.row_scale), x.grad.data_ptr(), x.c, stream),
              "cp_head1x1_bn_bwd_apply_f32(%s)" % self.name)
        x.has_grad = True
        if self.dgamma_p is not None or self.dbeta_p is not None:
            check(lib.cp_bn_param_grads_f32(self.red.data_ptr(), x.c, self.real_c, self.classes, _ptr(self.dgamma_p), _ptr(self.dbeta_p), stream),
                  "cp_bn_param_grads_f32(%s)" % self.name)

    def backward(self, stream: int):
        if self.head is not None:
            return self._backward_fused_head(stream)
        lib = _lib.load()
        x, y, p = self.x, self.y, self.plan
        C_ = x.c
        if not y.needs_grad:
            return  # bn_data: its beta gradient comes from conv0's weight gradient (TrainPlan.backward)
        assert y.has_grad, "gradient of %s output not produced" % self.name
        gam, bet = self.gamma_full.data_ptr(), self.beta_full.data_ptr()
        check(lib.cp_bn_act_bwd_reduce_f32(x.data.data_ptr(), C_, y.grad.data_ptr(), C_, x.pixels, C_, self.classes, self.mean.data_ptr(),
                                           self.rstd.data_ptr(), gam, bet, _ptr(self.labels), self.act, self.scale.data_ptr(), self.shift.data_ptr(),
                                           self.red.data_ptr(), self.chan.data_ptr(), stream), "cp_bn_act_bwd_reduce_f32(%s)" % self.name)
        n = p.all_reduce_stats(self.chan, x.pixels)
        if x.needs_grad:
            check(lib.cp_bn_act_bwd_apply_f32(x.data.data_ptr(), C_, y.grad.data_ptr(), C_, x.pixels, C_, self.mean.data_ptr(), self.rstd.data_ptr(),
                                              gam, bet, _ptr(self.labels), self.act, self.scale.data_ptr(), self.shift.data_ptr(), self.chan.data_ptr(),
                                              float(n), _ptr(self.row_scale),
                                              x.grad.data_ptr(), C_, 1 if x.has_grad else 0, stream), "cp_bn_act_bwd_apply_f32(%s)" % self.name)
            x.has_grad = True
        if self.dgamma_p is not None or self.dbeta_p is not None:
            check(lib.cp_bn_param_grads_f32(self.red.data_ptr(), C_, self.real_c, self.classes, _ptr(self.dgamma_p), _ptr(self.dbeta_p), stream),
                  "cp_bn_param_grads_f32(%s)" % self.name)


class FnOp:
    def __init__(self, fwd, bwd, reads=()):
        self.forward, self.backward = fwd, bwd
        self.reads = tuple(reads)   # tensors the closure reads (TrainPlan._consumers)


# ------------------------------------------------------------------------------------------------
# plan
# ------------------------------------------------------------------------------------------------
class TrainPlan:
    """Buffers + tape for one (batch, H, W); `group` = torch.distributed process group for SyncBN / DP (or None)."""

    GRAD_LD = 64   # loss gradient rows: [0,32) logits (seg_dim real), [32,64) vertex (ver_dim real)
    VERT_OFF = 32

    def __init__(self, store: ParamStore, seg_dim: int, ver_dim: int, batch: int, h: int, w: int,
                 decoder_dims: Sequence[int] = DECODER_DIMS_DEFAULT, group=None, world_size: int = 1,
                 partial: Sequence[bool] = PARTIAL_DEFAULT, guided: Sequence[bool] = GUIDED_DEFAULT, bilinear: Sequence[bool] = BILINEAR_DEFAULT,
                 pvnet: bool = False, shared: Sequence[bool] = (False,) * 5, reuse_first: bool = False, skips2: bool = True):
        if h % 8 or w % 8:
            raise ValueError("input height/width must be multiples of 8 (got %dx%d)" % (h, w))
        if seg_dim > 32 or (ver_dim > 32 and not pvnet):
            raise ValueError("the training plan supports up to 32 classes / 32 vertex channels (more vertex channels only for `pvnet` with separated vector fields)")
        lib = _lib.load()
        self.store, self.seg_dim, self.ver_dim = store, seg_dim, ver_dim
        self.batch, self.h, self.w = batch, h, w
        self.pvnet = bool(pvnet)
        if self.pvnet:  # one merged head: its gradient row is the contiguous [seg | vertex] record
            if seg_dim + ver_dim > self.GRAD_LD:
                # `pvnet` with SEPARATED vector fields (one 2*kp slice per object, train_casapose.py:57,97-125): a wider gradient row; a multiple
                # of 32 so that the head's data gradient may read its channel-padded rows
                self.GRAD_LD = (seg_dim + ver_dim + 31) // 32 * 32
            self.VERT_OFF = seg_dim
        self.group, self.world_size = group, world_size
        self._f16x2_mon = self._f16x2_host = self._f16x2_event = None   # range monitor of the f16x2 forward (_poll_f16x2)
        self._f16x2_steps, self.f16x2_checks, self.f16x2_demoted = 0, 0, []
        self._bwd_f16, self._bwd_calibrated, self.f16x2_bwd_moves = None, False, []   # backward GEMMs in f16x2 (train_bwd_f16x2)
        self.loss_exp, self._dout_scale = 0, 1.0                                       # power of two on the loss (direct data gradients in f16x2)
        self.comm_timing = None   # start_comm_timing()
        self.comm_log = None      # start_comm_log()
        self._buckets = None
        self._pending: List = []
        self.update_moving = True
        dev = store.device
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        B, K, V = batch, seg_dim, ver_dim
        self.out_ld = (K + V + 3) // 4 * 4   # row length of the output record, padded to 16 bytes (the LS voter stages rows as float4)
        hs = [h, h // 2, h // 4, h // 8]
        ws = [w, w // 2, w // 4, w // 8]
        self.ops: List = []
        self.convs: List[TrainConv] = []
        store.pack_reset()  # this plan's kernel layouts share one arena (one gather per weight refresh)
        dims = tuple(decoder_dims)

        def new(hh, ww, c, grad=True, name=""):
            return TT(torch.empty(B, hh, ww, c, **f32), grad, name)

        self.img4 = new(h, w, 4, False, "img4")        # raw image, channel 3 = 0 (decoder skip)
        self.x0 = new(h, w, 4, False, "bn_data")        # bn_data(image), channel 3 = 1 (see below)
        self.out = torch.zeros(B, h, w, self.out_ld, **f32)         # padded record
        self.out_view = self.out[..., :K + V]                       # what the model returns: concat(seg logits, vertex)
        self.dout = torch.zeros(B, h, w, self.GRAD_LD, **f32)
        self.labels = [torch.empty(B, hs[l], ws[l], **u8) for l in range(4)]
        self.pnorm = [torch.empty(B, hs[l], ws[l], **f32) for l in range(4)]
        self.sel = [torch.empty(B, hs[l], ws[l], **u8) for l in range(3)]
        self.sel_zero = [torch.zeros(B, hs[l], ws[l], **u8) for l in range(3)]  # plain nearest x2
        self.partial, self.guided = tuple(bool(v) for v in partial), tuple(bool(v) for v in guided)
        self.bilinear = tuple(bool(v) for v in bilinear)
        # weight sharing between the decoders (the `_sw*` registry entries; see engine.CasaposeNet)
        self.shared, self.reuse_first, self.skips2 = tuple(bool(v) for v in shared), bool(reuse_first), bool(skips2)
        self.gmask = [torch.empty(B, hs[l], ws[l], **u8) if any(self.bilinear) else None for l in range(3)]
        self.loss_sums = torch.zeros(3, dtype=torch.float64, device=dev)
        self.loss_ws = torch.empty(max(lib.cp_pose_loss_workspace_bytes(B, h, w), lib.cp_pose_loss_sep_workspace_bytes(B, h, w, K)), **u8)
        self.object_loss_values = torch.zeros(B, K - 1, **f32)  # per-object proxy distances (proxy_voting_dist)
        # keypoint-reprojection loss (LS voter forward/backward)
        oc, kp = K - 1, 9
        self.est_labels = torch.empty(B, h, w, **u8)
        self.kp_counts = torch.zeros(2, B, K, dtype=torch.int32, device=dev)
        self.kp_conf_sums = torch.zeros(B, kp, dtype=torch.float64, device=dev)
        self.ls_sums = torch.zeros(B * oc * kp * 5, dtype=torch.float64, device=dev)
        self.ls_pu = torch.zeros(B * oc * kp * 4, **f32)
        self.ls_coords = torch.zeros(B, oc, kp, 2, **f32)
        self.ls_g = torch.zeros(B, oc, kp, 2, **f32)
        self.kp_loss_val = torch.zeros(1, dtype=torch.float64, device=dev)

        def layer(key, layout, k, cout, sources, grad_sources):
            L = TrainConv(store, key, layout, k, cout, sources, grad_sources)
            self.convs.append(L)
            return L

        def conv(L, srcs, o: TT, in_h, in_w, **kw):
            op = ConvOp(L, srcs, (o.data, 0, o.c), B, in_h, in_w, out=o, **kw)
            self.ops.append(op)
            return op

        def bn(name, x, y, act, gamma=True, labels=None, classes=1, row_scale=None, pad_one=False, clade=False):
            gk = (name + ".gamma") if gamma else None
            self.ops.append(BnActOp(self, name, x, y, act, gk, name + ".beta", labels, classes, row_scale, pad_one))

        RELU, LEAKY, NONE = _lib.ACT_RELU, _lib.ACT_LEAKY01, _lib.ACT_NONE
        # ---- encoder --------------------------------------------------------------------------------
        # bn_data has no gamma; its padding channel is forced to the constant 1 so that conv0's weight gradient
        # for that (zero-weight) channel is G[t][o] = sum of dy over the positions where tap t is inside the image:
        # d beta_data[c] = sum_{t,o} W0[t,c,o] * G[t][o] without a 7x7 transposed convolution for three numbers.
        bn("bn_data", self.img4, self.x0, NONE, gamma=False, pad_one=True)
        self.bn_data_op = self.ops[-1]
        c0 = layer("conv0.kernel", 0, 7, 64, [(4, 3)], [False])
        self.conv0 = c0
        x = new(hs[1], ws[1], 64)
        conv(c0, [(self.x0, 4)], x, h, w, stride=2, pad=3)
        x2s = new(hs[1], ws[1], 64, name="x2s")
        bn("bn0", x, x2s, RELU)
        pooled = new(hs[2], ws[2], 64, name="pool")

        pool_idx = torch.empty(B, hs[2], ws[2], 64, dtype=torch.uint8, device=dev)   # arg-max tap per pooled element: the adjoint routes by it

        def pool_f(stream, src=x2s, dst=pooled):
            check(lib.cp_maxpool3x3s2_idx_f32(src.data.data_ptr(), B, hs[1], ws[1], 64, dst.data.data_ptr(), pool_idx.data_ptr(), stream), "cp_maxpool3x3s2_idx_f32")

        def pool_b(stream, src=x2s, dst=pooled):
            assert dst.has_grad
            check(lib.cp_maxpool3x3s2_bwd_idx_f32(pool_idx.data_ptr(), dst.grad.data_ptr(), B, hs[1], ws[1], 64, src.grad.data_ptr(),
                                                  1 if src.has_grad else 0, stream), "cp_maxpool3x3s2_bwd_idx_f32")
            src.has_grad = True

        self.ops.append(FnOp(pool_f, pool_b, reads=[x2s]))
        xr, cur_h, cur_w, cin = pooled, hs[2], ws[2], 64
        taps: Dict[str, TT] = {"x2s": x2s}
        tap_names = ["x4s", "x8s", "x16s", "x32s"]
        for s, f in enumerate(STAGE_FILTERS):
            dl = STAGE_DILATION[s]
            for u in range(2):
                base = "stage%d_unit%d_" % (s + 1, u + 1)
                stride = STAGE_STRIDE[s] if u == 0 else 1
                oh, ow = (cur_h - 1) // stride + 1, (cur_w - 1) // stride + 1
                a = new(cur_h, cur_w, cin, name=base + "a")
                bn(base + "bn1", xr, a, RELU)
                if u == 0:
                    if s > 0:
                        taps[tap_names[s - 1]] = a
                    sc = new(oh, ow, f, name=base + "sc")
                    sc_layer = layer(base + "sc.kernel", 0, 1, f, [(cin, cin)], [True])
                    shortcut = sc
                else:
                    shortcut = xr
                t = new(oh, ow, f)
                conv(layer(base + "conv1.kernel", 0, 3, f, [(cin, cin)], [True]), [(a, cin)], t, cur_h, cur_w, stride=stride, dilation=dl, pad=dl)
                if u == 0:
                    # the shortcut comes AFTER conv1 in the tape (both read `a`): the backward then runs its data gradient first, which lets the
                    # GEMM route write `a.grad` plainly while conv1's data gradient, running after it, accumulates through its residual input
                    conv(sc_layer, [(a, cin)], sc, cur_h, cur_w, stride=stride)
                t2 = new(oh, ow, f)
                bn(base + "bn2", t, t2, RELU)
                xn = new(oh, ow, f, name=base + "out")
                conv(layer(base + "conv2.kernel", 0, 3, f, [(f, f)], [True]), [(t2, f)], xn, oh, ow, dilation=dl, pad=dl, residual=shortcut)
                xr, cur_h, cur_w, cin = xn, oh, ow, f
        x32s = new(cur_h, cur_w, 512, name="x32s")
        bn("bn1", xr, x32s, RELU)
        taps["x32s"] = x32s
        self.taps = taps
        skips = [None, taps["x8s"], taps["x4s"], taps["x2s"], self.img4]
        skip_c = [None, (128, 128), (64, 64), (64, 64), (4, 3)]
        lvl = [3, 3, 2, 1, 0]

        def upsample(prev: TT, l: int, guided: bool, selmap: Optional[torch.Tensor] = None, blend: bool = False) -> TT:
            big = new(hs[l], ws[l], prev.c)
            sh, sw, c = hs[l] // 2, ws[l] // 2, prev.c
            selmap = self.sel[l] if selmap is None else selmap

            def f(stream):
                if blend:
                    check(lib.cp_guided_bilinear_upsample_x2_f32(prev.data.data_ptr(), self.gmask[l].data_ptr(), B, sh, sw, c, big.data.data_ptr(), stream),
                          "cp_guided_bilinear_upsample_x2_f32")
                elif guided:
                    check(lib.cp_guided_upsample_x2_f32(prev.data.data_ptr(), selmap.data_ptr(), B, sh, sw, c, big.data.data_ptr(), stream), "cp_guided_upsample_x2_f32")
                else:
                    check(lib.cp_upsample_bilinear_x2_f32(prev.data.data_ptr(), B, sh, sw, c, big.data.data_ptr(), stream), "cp_upsample_bilinear_x2_f32")

            def b(stream):
                assert big.has_grad and not prev.has_grad
                if blend:
                    check(lib.cp_guided_bilinear_upsample_x2_bwd_f32(big.grad.data_ptr(), c, self.gmask[l].data_ptr(), B, sh, sw, c, prev.grad.data_ptr(), stream),
                          "cp_guided_bilinear_upsample_x2_bwd_f32")
                elif guided:
                    check(lib.cp_guided_upsample_x2_bwd_f32(big.grad.data_ptr(), c, selmap.data_ptr(), B, sh, sw, c, prev.grad.data_ptr(), stream), "cp_guided_upsample_x2_bwd_f32")
                else:
                    check(lib.cp_upsample_bilinear_x2_bwd_f32(big.grad.data_ptr(), c, B, sh, sw, c, prev.grad.data_ptr(), stream), "cp_upsample_bilinear_x2_bwd_f32")
                prev.has_grad = True

            self.ops.append(FnOp(f, b, reads=[prev]))
            return big

        def decoder(first: int, second: bool):
            prev = None
            for i in range(5):
                l = lvl[i]
                idx = first + i
                partial = second and self.partial[i]
                act_kind = RELU if i == 0 else LEAKY
                if second and i == 0 and self.reuse_first:  # casa_layer(y, "6", skip_conv=True): CLADE on block 1's raw convolution output
                    act = new(hs[l], ws[l], dims[i])
                    bn("pv_block_%d_clade" % idx, self._y_raw, act, act_kind, labels=self.labels[l], classes=K)
                    prev = act
                    continue
                if self.shared[i]:  # one PartialConvolution weight set for blocks i+1 and i+6 ([Cin,3,3,Cout])
                    key, layout = "pv_block_%d_%d_conv2d.weights" % (i + 1, i + 6), 1
                elif partial:
                    key, layout = "pv_block_%d_prepare_conv2d.weights" % idx, 1
                else:
                    key, layout = "pv_block_%d_conv2d.kernel" % idx, 0
                if i == 0:
                    srcs, tts, gs = [(512, 512)], [(x32s, 512)], [True]
                else:
                    if i >= 2:  # the previous block upsampled its output: bilinear (decoder 1), label-guided or plain nearest (decoder 2)
                        src0 = upsample(prev, l, second, None if (not second or self.guided[i - 1]) else self.sel_zero[l],
                                        blend=second and self.guided[i - 1] and self.bilinear[i - 1])
                    else:
                        src0 = prev
                    srcs = [(dims[i - 1], dims[i - 1]), skip_c[i]]
                    tts = [(src0, dims[i - 1]), (skips[i], skips[i].c)]
                    gs = [True, skips[i].needs_grad]
                    if second and not self.skips2:
                        srcs, tts, gs = srcs[:1], tts[:1], gs[:1]
                L = layer(key, layout, 3, dims[i], srcs, gs)
                raw = new(hs[l], ws[l], dims[i])
                act = new(hs[l], ws[l], dims[i])
                if not second and i == 0:
                    self._y_raw = raw
                if partial:
                    conv(L, tts, raw, hs[l], ws[l], pad=1, tap_label=self.labels[l], row_scale=self.pnorm[l])
                    bn("pv_block_%d_clade" % idx, raw, act, act_kind, labels=self.labels[l], classes=K, row_scale=self.pnorm[l])
                elif second:
                    conv(L, tts, raw, hs[l], ws[l], pad=1)
                    bn("pv_block_%d_clade" % idx, raw, act, act_kind, labels=self.labels[l], classes=K)
                else:
                    conv(L, tts, raw, hs[l], ws[l], pad=1)
                    bn("pv_block_%d_bn" % idx, raw, act, act_kind)
                prev = act
            return prev

        feat1 = decoder(1, False)
        self.cond_labels = None
        if self.pvnet:  # PVNet (pose_models.py:645-696)
            head = layer("pv_final_conv.kernel", 0, 1, K + V, [(dims[4], dims[4])], [True])
            self.ops.append(ConvOp(head, [(feat1, dims[4])], (self.out, 0, self.out_ld), B, h, w, out=None, dy_ptr_ld=(self.dout, 0, self.GRAD_LD)))
            self._finish_plan(f32)
            return
        seg_head = layer("pv_final_conv_segmentation.kernel", 0, 1, K, [(dims[4], dims[4])], [True])
        self.ops.append(ConvOp(seg_head, [(feat1, dims[4])], (self.out, 0, self.out_ld), B, h, w, out=None, dy_ptr_ld=(self.dout, 0, self.GRAD_LD)))
        self.cond_labels: Optional[torch.Tensor] = None  # ground-truth conditioning (train_vectors_with_ground_truth)

        def label_f(stream):
            if self.cond_labels is not None:
                self.labels[0].copy_(self.cond_labels)
            else:
                logits, ld_ = (self.seg_dense, K) if getattr(self, "seg_dense", None) is not None else (self.out, self.out_ld)
                check(lib.cp_argmax_labels(logits.data_ptr(), ld_, K, B * h * w, self.labels[0].data_ptr(), stream), "cp_argmax_labels")
            lab = (C.c_void_p * 4)(*[t.data_ptr() for t in self.labels])
            pn = (C.c_void_p * 4)(*[t.data_ptr() for t in self.pnorm])
            sl = (C.c_void_p * 3)(*[t.data_ptr() for t in self.sel])
            check(lib.cp_label_pyramid(self.labels[0].data_ptr(), B, h, w, lab, pn, sl, stream), "cp_label_pyramid")
            for l_ in range(3):
                if self.gmask[l_] is not None:
                    check(lib.cp_guided_match_mask(self.labels[l_].data_ptr(), self.labels[l_ + 1].data_ptr(), B, hs[l_], ws[l_], self.gmask[l_].data_ptr(), stream),
                          "cp_guided_match_mask")

        self.ops.append(FnOp(label_f, lambda stream: None))
        feat2 = decoder(6, True)
        ver_head = layer("pv_final_conv_vertex.kernel", 0, 1, V, [(dims[4], dims[4])], [True])
        self.ops.append(ConvOp(ver_head, [(feat2, dims[4])], (self.out, K, self.out_ld), B, h, w, out=None, dy_ptr_ld=(self.dout, self.VERT_OFF, self.GRAD_LD)))
        self._finish_plan(f32)

    def _finish_plan(self, f32):
        dev = self.store.device
        c0 = self.conv0
        self.store.pack_finalize()
        # two convolutions on one weight set: the op that runs LAST in the backward (first in the forward) adds to the master gradient
        seen = set()
        for op in reversed(self.ops):
            if isinstance(op, ConvOp):
                op.accumulate_master = op.layer.key in seen
                seen.add(op.layer.key)
        self.tensors = [o for o in self._all_tensors()]
        # fused training normalisation (CASAPOSE_FUSE_NORM=0 restores the separate passes everywhere)
        fuse = os.environ.get("CASAPOSE_FUSE_NORM", "1")   # 1 = all, 0 = none, heads / wino = one family only (bisection aid)
        self.fuse_norm = fuse != "0"
        self._fuse_wino_ok = fuse in ("1", "wino", "wino_stats", "wino_pre")
        self._fuse_mode = fuse
        if fuse in ("1", "heads"):
            self._fuse_heads()
        # weight gradients on a second stream (CASAPOSE_WGRAD_STREAM=1; see backward())
        self.wgrad_on_side_stream = os.environ.get("CASAPOSE_WGRAD_STREAM", "0") == "1"
        self._side = torch.cuda.Stream(device=dev) if self.wgrad_on_side_stream else None
        for op in self.ops:
            if isinstance(op, ConvOp):
                op.setup_gemm()
        # Winograd for the deep 3x3 layers (forward and data gradient); shared scratch sized for the largest of them
        self.use_winograd = os.environ.get("CASAPOSE_NO_WINOGRAD", "0") != "1"
        self.wino_V = self.wino_M = None
        if self.use_winograd:
            sizes = [(op, op.setup_winograd()) for op in self.ops if isinstance(op, ConvOp)]
            nv = max([sz[0] for _, sz in sizes] + [0])
            nm = max([sz[1] for _, sz in sizes] + [0])
            if nv:
                self.wino_V = torch.empty(nv, **f32)
                self.wino_M = torch.empty(nm, **f32)
                self.wino_M2 = torch.empty(nm, **f32) if self.wgrad_on_side_stream else None   # dY planes of the side stream's weight gradients
                for op, sz in sizes:
                    if sz[0]:
                        op.bind_winograd(self.wino_V, self.wino_M, self.wino_M2)
                if self._fuse_wino_ok:
                    self._fuse_winograd()
        # gather map of conv0's packed weight-gradient entries that belong to the padding channel (c = 3)
        ramp = np.zeros((7, 7, 3, 64), np.int64)  # unused values; only the layout matters
        k0 = np.full((64, c0.ktot), -1, np.int64)
        for t in range(49):
            k0[:, t * 4 + 3] = np.arange(64) * c0.ktot + t * 4 + 3
        self.g_idx = torch.from_numpy(k0[:, [t * 4 + 3 for t in range(49)]].T.copy()).to(dev)  # [49 taps][64 cout] -> flat index into dwp
        del ramp

    def _consumers(self, t: TT) -> int:
        """ops that READ tensor t (convolution sources / residuals, normalisation inputs, the pooling / resampling closures); taps are
        checked by the caller."""
        n = 0
        for op in self.ops:
            if isinstance(op, ConvOp):
                n += sum(1 for s, _ in op.srcs if s is t) + (1 if op.residual is t else 0)
            elif isinstance(op, BnActOp):
                n += 1 if op.x is t else 0
            elif isinstance(op, FnOp):
                n += sum(1 for r in op.reads if r is t)
        return n

    def _fuse_heads(self):
        """Blocks 5 / 10 -> head: the normalisation op directly before a streaming 1x1 head whose activated output nobody else reads hands its
        tables to the head (forward, weight gradient) and recomputes the head's data gradient in its own backward (csrc/head1x1.hip)."""
        for i, op in enumerate(self.ops):
            if not (isinstance(op, ConvOp) and op.head_fast and i > 0 and isinstance(self.ops[i - 1], BnActOp)):
                continue
            bn = self.ops[i - 1]
            y = op.srcs[0][0]
            if bn.y is not y or bn.x.c != 32 or bn.x.pixels % 32 or self._consumers(y) != 1 or any(t is y for t in self.taps.values()):
                continue
            if op.dy_ptr_ld is None or not bn.x.needs_grad or bn.classes > 64:
                continue
            st_, off, ld = op.dy_ptr_ld
            if ld % 4 or off % 4 or ld - off % ld < 32:
                continue
            bn.head, op.pre_bn = op, bn
        self._whole_records()

    def _whole_records(self):
        """Both fused heads write slices of the same [pixels][K + V] records, and a head that fills 36 or 108 bytes of every 144 leaves each
        128-byte line partly written: the memory system answers with a read-modify-write (556 us for the 9-column head into the records against
        209 us into dense rows, tools/debug/head_probe.py).  So the segmentation head writes DENSE rows of K logits (self.seg_dense; the arg-max
        reads those) and the vertex head, the last writer, copies them in front of its own columns: cp_head1x1_fwd_affine_record_f32 writes
        complete records.  CASAPOSE_HEAD_RECORDS=0 keeps the two slice writers."""
        self.seg_dense = None
        K, V = self.seg_dim, self.ver_dim
        heads = [op for op in self.ops if isinstance(op, ConvOp) and op.head_fast and op.pre_bn is not None]
        if os.environ.get("CASAPOSE_HEAD_RECORDS", "1") == "0" or len(heads) != 2 or K > 16 or K + V != self.out_ld:
            return
        seg, ver = heads
        if seg._out_ptr_ld != (self.out, 0, self.out_ld) or ver._out_ptr_ld != (self.out, K, self.out_ld) or seg.layer.cout != K or ver.layer.cout != V:
            return
        self.seg_dense = torch.zeros(self.out.shape[0] * self.out.shape[1] * self.out.shape[2], K, dtype=torch.float32, device=self.out.device)
        seg._out_ptr_ld = (self.seg_dense, 0, K)
        ver.record_prefix = (self.seg_dense, K, K)

    def _fuse_winograd(self):
        """Normalisation layers next to Winograd convolutions: the producer's output transform owns (tile, 4 channels) per lane and accumulates
        the batch statistics on its way out (no cp_bn_stats_f32 pass); a normalisation layer without labels whose activated output is read by ONE
        Winograd convolution only hands its tables to that convolution's input transform (no cp_affine_act_f32 pass, y never written -- the
        weight gradient of a Winograd layer multiplies the kept transformed input V, not y)."""
        convs = [op for op in self.ops if isinstance(op, ConvOp)]
        for bn in [op for op in self.ops if isinstance(op, BnActOp)]:
            if bn.head is not None or bn.pad_one:
                continue
            prod = [c for c in convs if c.out is bn.x and getattr(c, "wino_fwd", None) is not None and c.stats_to is None]
            if prod and bn.x.c == prod[0].layer.cout and self._fuse_mode != "wino_pre":
                prod[0].stats_to, bn.stats_from = bn, prod[0]
            if self._fuse_mode == "wino_stats":
                continue
            if bn.classes != 1 or bn.labels is not None or any(t is bn.y for t in self.taps.values()) or self._consumers(bn.y) != 1:
                continue
            for c in convs:
                for i, (t, ld) in enumerate(c.srcs):
                    if t is bn.y and getattr(c, "wino_fwd", None) is not None and ld == bn.y.c and c.layer.sources[i][0] == bn.y.c:
                        c.pre_norm[i], bn.consumer = bn, c

    def _all_tensors(self):
        seen = {}
        for op in self.ops:
            for attr in ("x", "y", "out", "residual"):
                t = getattr(op, attr, None)
                if isinstance(t, TT):
                    seen[id(t)] = t
            for t, _ in getattr(op, "srcs", []):
                seen[id(t)] = t
        for t in self.taps.values():
            seen[id(t)] = t
        return list(seen.values())

    def activation_pattern(self) -> Dict[str, torch.Tensor]:
        """{normalisation layer name: bool tensor} -- the branch every ReLU / leaky pair took in the last forward (the sign of its output).
        The backward differentiates exactly this piecewise-linear function (cp_bn_act_bwd_* decide the branch with the forward's own
        expression); the gradient tests hand the pattern to the fp64 oracle so that elements sitting within rounding of a kink do not
        count as gradient error."""
        stream = torch.cuda.current_stream(self.store.device).cuda_stream
        for op in self.ops:
            if isinstance(op, BnActOp) and (op.head is not None or op.consumer is not None):
                op.materialize(stream)   # a fused layer never stored y: evaluate the forward's expression once for the pattern
        return {op.name: (op.y.data > 0).cpu() for op in self.ops if isinstance(op, BnActOp) and op.act != _lib.ACT_NONE}

    # ---- distributed hooks ---------------------------------------------------------------------------
    def all_reduce_stats(self, table: torch.Tensor, local_pixels: int) -> int:
        """SUM the fp64 statistic table over the replicas; returns the global pixel count."""
        if self.comm_log is not None:   # structure of the step's exchanges (comm_structure()): recorded with or without replicas
            self.comm_log.append(("syncbn", table.numel() * table.element_size(), "blocking", "compute"))
        if self.comm_timing is not None and self.group is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            parallel.all_reduce_sum_(table, self.group, self.world_size)
            e1.record()
            self.comm_timing["syncbn"].append((e0, e1))
        else:
            parallel.all_reduce_sum_(table, self.group, self.world_size)
        return local_pixels * self.world_size

    # ---- structure of the step's exchanges (round 6): what is launched, in which order, on which stream ---------------------------------
    def start_comm_log(self):
        """From the next step on, record every exchange point of a step in launch order: ("syncbn", bytes, "blocking", "compute") for a statistic
        table all-reduce (the next kernel needs it), ("op", index) for every backward op, ("grad_bucket", bytes, "async", stream id of the
        compute stream at launch, first op index) where a gradient bucket's all-reduce is launched, ("grad_wait", n) where the compute stream
        waits for the buckets.  Recorded with or without replicas (a single replica launches no collective but passes the same points), so that
        the overlap of the data-parallel step can be asserted structurally (tests/test_gpu_dp.py) and counted (bench.py --mode train)."""
        self.comm_log = []

    def comm_structure(self) -> dict:
        """Per step, from the log of the LAST logged step: blocking collectives (SyncBN tables) and their payload, gradient buckets, their payload and
        how many backward ops are launched AFTER each bucket's all-reduce (what its exchange can hide behind)."""
        log = self.comm_log or []
        last = len(log) - 1 - next((i for i, e in enumerate(reversed(log)) if e[0] == "step_begin"), len(log) - 1)
        step = log[last + 1:] if log and log[last][0] == "step_begin" else log
        bn = [e for e in step if e[0] == "syncbn"]
        buckets, after = [], []
        for i, e in enumerate(step):
            if e[0] == "grad_bucket":
                buckets.append(e)
                after.append(sum(1 for x in step[i + 1:] if x[0] == "op"))
        return {"blocking_collectives_per_step": len(bn), "blocking_payload_bytes_per_step": int(sum(e[1] for e in bn)),
                "gradient_buckets": len(buckets), "gradient_payload_bytes_per_step": int(sum(e[1] for e in buckets)),
                "backward_ops_launched_after_each_bucket": after, "backward_ops": sum(1 for e in step if e[0] == "op"),
                "note": "structure of the data-parallel step, identical for every world size: the blocking calls sit on the critical path (global-batch "
                        "statistics, as the reference's SyncBatchNormalization), each gradient bucket's all-reduce is launched asynchronously when the "
                        "backward has passed the bucket's first layer"}

    # ---- communication accounting (bench.py --mode train with N > 1 ranks; tests/test_gpu_dp.py) ------------------
    def start_comm_timing(self):
        """From the next step on, bracket every collective the step WAITS for with events on the compute stream: the 58 SyncBN table
        all-reduces (blocking: the next kernel needs the global statistics) and the wait for the gradient buckets in all_reduce_grads()
        (whatever of their exchange the backward did not cover).  comm_report() turns them into milliseconds per step."""
        self.comm_timing = {"syncbn": [], "grad_wait": [], "steps": 0}

    def comm_report(self) -> dict:
        """{"syncbn_ms", "syncbn_calls", "grad_wait_ms", "exposed_ms", "grad_total_ms", "grad_hidden_ms", "grad_bytes"} per step.  exposed =
        time the compute stream spent inside / waiting for collectives; grad_total = the same gradient buckets all-reduced back to back on an
        idle GPU (measured here, after the steps), so grad_hidden = grad_total - grad_wait is what the overlap with the backward bought."""
        t = self.comm_timing
        torch.cuda.synchronize(self.store.device)
        steps = max(t["steps"], 1)
        bn = sum(a.elapsed_time(b) for a, b in t["syncbn"]) / steps
        gw = sum(a.elapsed_time(b) for a, b in t["grad_wait"]) / steps
        total, nbytes = 0.0, 0
        if self.group is not None and self._buckets:
            scratch = torch.zeros_like(self.store.grad)
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                hs = [parallel.all_reduce_sum_async(scratch[a:e], self.group) for _, a, e in self._buckets]
                for h in hs:
                    h.wait()
                e1.record()
                e1.synchronize()
                total = e0.elapsed_time(e1)   # the last repetition (the first pays RCCL's lazy set-up)
            nbytes = 4 * scratch.numel()
        return {"syncbn_ms": round(bn, 4), "syncbn_calls": len(t["syncbn"]) // steps, "grad_wait_ms": round(gw, 4), "exposed_ms": round(bn + gw, 4),
                "grad_total_ms": round(total, 4), "grad_hidden_ms": round(max(total - gw, 0.0), 4), "grad_bytes": nbytes}

    # ---- one step ------------------------------------------------------------------------------------
    def refresh_weights(self, stream: int):
        self.store.pack_refresh(stream)
        for L in self.convs:
            L.refresh(stream, hooks_only=True)

    def forward(self, img: torch.Tensor, cond_labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        lib = _lib.load()
        B, h, w = self.batch, self.h, self.w
        if tuple(img.shape) != (B, h, w, 3) or img.dtype != torch.float32 or not img.is_contiguous():
            raise ValueError("image must be a contiguous float32 [%d,%d,%d,3] tensor" % (B, h, w))
        stream = torch.cuda.current_stream(img.device).cuda_stream
        self.cond_labels = cond_labels
        check(lib.cp_pad_channels_3to4(img.data_ptr(), self.img4.data.data_ptr(), B * h * w, stream), "cp_pad_channels_3to4")
        self._poll_f16x2(stream)
        for op in self.ops:
            op.forward(stream)
        self._read_f16x2(img.device)
        return self.out_view

    # ---- range monitor of the f16x2 forward (round 6) -------------------------------------------------------------------------------
    def _bwd_slots(self):
        """[(op, state, entry)] of the backward GEMMs that run (or will run) in the fp16 two-way split: Winograd data gradients (entry = the
        wino_dgrad dict, its weights are re-packed when the exponent moves) and Winograd weight gradients (entry None)"""
        if self._bwd_f16 is None:
            self._bwd_f16 = []
            for op in self.ops:
                if not isinstance(op, ConvOp):
                    continue
                self._bwd_f16 += [(op, w["f16"], w) for w in getattr(op, "wino_dgrad", {}).values() if w.get("f16") is not None]
                wf = getattr(op, "wino_fwd", None)
                if wf is not None and wf.get("wg16") is not None:
                    self._bwd_f16.append((op, wf["wg16"], None))
                if op.direct_dgrad_split() or (train_bwd_f16x2() and op.wgrad_planes() == 3):
                    # direct 3x3 data / weight gradients (conv_hsplit, conv_wgrad_split): entry "direct", state on / off instead of an exponent
                    op.bw16 = dict(on=False, mon=None, dead=False, e=None)
                    self._bwd_f16.append((op, op.bw16, "direct"))
        return self._bwd_f16

    @staticmethod
    def _set_bwd_exponent(op, f, entry, e, stream):
        if entry is not None:
            op.set_dgrad_exponent(entry, e, stream)
        else:
            f["e"] = e

    def _set_loss_exponent(self, e_new: int, stream: int):
        """move the power of two on the loss; the Winograd GEMMs' own exponents move the other way at the same moment (their operands carry it)"""
        d = e_new - self.loss_exp
        if d == 0:
            return
        self.loss_exp = e_new
        for op, f, entry in self._bwd_slots():
            if entry != "direct" and f["e"] is not None and not f["dead"]:
                self._set_bwd_exponent(op, f, entry, f["e"] - d, stream)

    def _judge_direct(self, vals, stream: int):
        """vals: {slot index j: max |dY| as measured, i.e. including the loss factor in force}.  Moves the loss exponent when the largest of them
        has left [2^7, 2^13), then switches every direct data gradient on (inside [1, 2^13]) or off (outside [0.25, 2^14])."""
        bwd = self._bwd_slots()
        live = {j: v for j, v in vals.items() if np.isfinite(v) and v > 0.0}
        if not live:
            return
        top = max(live.values())
        shift = 0
        if not (2.0 ** 7 <= top < 2.0 ** 13):
            shift = 10 - int(np.floor(np.log2(top)))
            self._set_loss_exponent(int(np.clip(self.loss_exp + shift, -100, 100)), stream)
        for j, v in vals.items():
            op, f, _ = bwd[j]
            if f["dead"]:
                continue
            if not np.isfinite(v):
                f["dead"] = True
                op.set_direct_dgrad_f16x2(False, stream)
                continue
            v2 = v * 2.0 ** shift
            if not f["on"] and 1.0 <= v2 <= 2.0 ** 13:
                op.set_direct_dgrad_f16x2(True, stream)
            elif f["on"] and not (0.25 <= v2 <= 2.0 ** 14):
                op.set_direct_dgrad_f16x2(False, stream)

    def _arm_f16x2(self):
        """one monitor slot per convolution op whose forward runs in the fp16 two-way split (slot i <-> self.ops[i]), then one per Winograd data
        gradient of train_bwd_f16x2()"""
        bwd = self._bwd_slots()
        if self._f16x2_mon is None:
            dev = self.out.device
            self._f16x2_mon = torch.zeros(4 * (len(self.ops) + len(bwd)), dtype=torch.int32, device=dev)
            self._f16x2_host = torch.zeros(4 * (len(self.ops) + len(bwd)), dtype=torch.int32).pin_memory()
        base = self._f16x2_mon.data_ptr()
        for i, op in enumerate(self.ops):
            if isinstance(op, ConvOp):
                op.mon_ptr = base + 16 * i if op.layer.fwd_f16x2 else None
        for j, (_, f, _e) in enumerate(bwd):
            f["mon"] = base + 16 * (len(self.ops) + j)

    def _calibrate_bwd(self, stream: int):
        """after the plan's FIRST backward (exact split, slots armed): one synchronous reading gives every Winograd data gradient its exponent"""
        self._bwd_calibrated = True
        bwd = self._bwd_slots()
        if not bwd or self._f16x2_mon is None:
            return
        n0 = len(self.ops)
        w32 = self._f16x2_mon[4 * n0:].cpu().numpy().view(np.uint32).reshape(-1, 4)
        self._f16x2_mon[4 * n0:].zero_()
        direct = {}
        for j, (op, f, entry) in enumerate(bwd):
            if int(w32[j, 1]) == 0 or f["e"] is not None:
                continue
            amax = float(w32[j, :1].view(np.float32)[0])
            if entry == "direct":
                direct[j] = amax
            elif np.isfinite(amax) and amax > 0.0:
                self._set_bwd_exponent(op, f, entry, int(np.clip(10 - int(np.floor(np.log2(amax))), -100, 100)), stream)
        self._judge_direct(direct, stream)   # (moves the Winograd exponents just set by the loss exponent it chooses)

    def _read_f16x2(self, dev):
        """every F16X2_TRAIN_CHECK_EVERY-th step: the slots (sticky maxima over the steps since the last reading) travel to pinned host memory, are
        zeroed behind the copy and judged at the start of a later step -- no synchronisation in the step"""
        if self._f16x2_mon is None:
            return
        self._f16x2_steps += 1
        if self._f16x2_steps >= F16X2_TRAIN_CHECK_EVERY and self._f16x2_event is None:
            self._f16x2_host.copy_(self._f16x2_mon, non_blocking=True)
            self._f16x2_mon.zero_()
            self._f16x2_event = torch.cuda.Event()
            self._f16x2_event.record(torch.cuda.current_stream(dev))
            self._f16x2_steps = 0

    def _poll_f16x2(self, stream: int):
        from . import engine

        if self._f16x2_mon is None:
            if any(isinstance(op, ConvOp) and op.layer.fwd_f16x2 for op in self.ops) or self._bwd_slots():
                self._arm_f16x2()
            return
        ev = self._f16x2_event
        if ev is None or not ev.query():
            return
        self._f16x2_event = None
        lib = _lib.load()
        lo, hi = engine.F16X2_AMAX_LO / engine.F16X2_MONITOR_SLACK, engine.F16X2_AMAX_HI * engine.F16X2_MONITOR_SLACK
        w = self._f16x2_host.numpy().copy().view(np.uint32).reshape(-1, 4)
        out = []
        for i, op in enumerate(self.ops):
            if not isinstance(op, ConvOp) or not op.layer.fwd_f16x2 or int(w[i, 1]) == 0:
                continue
            amax = float(w[i, :1].view(np.float32)[0])
            if lib.cp_f16x2_range_check(amax, lo, hi, None) != 0:
                op.demote_forward_to_exact_split(stream)
                op.mon_ptr = None
                out.append("%s (max %.3g)" % (op.layer.name, amax))
        n0 = len(self.ops)
        direct = {}
        for j, (op, f, entry) in enumerate(self._bwd_slots()):
            # backward GEMMs: a data gradient's slot holds max |V 2^e| (the transform applies the factor), a weight gradient's max |dM| (the GEMM
            # applies it); e follows a drift out of [2^7, 2^13), a non-finite maximum ends the f16x2 run of that GEMM
            if f["dead"] or int(w[n0 + j, 1]) == 0:
                continue
            amax = float(w[n0 + j, :1].view(np.float32)[0])
            if entry == "direct":
                direct[j] = amax
                continue
            scaled = amax * (2.0 ** f["e"] if (entry is None and f["e"] is not None) else 1.0)
            if not np.isfinite(amax):
                f["dead"] = True
                self._set_bwd_exponent(op, f, entry, None, stream)
                out.append("%s %s gradient (max %.3g)" % (op.layer.name, "data" if entry is not None else "weight", amax))
            elif f["e"] is None:
                if amax > 0.0:
                    self._set_bwd_exponent(op, f, entry, int(np.clip(10 - int(np.floor(np.log2(amax))), -100, 100)), stream)
            elif scaled > 0.0 and not (2.0 ** 7 <= scaled < 2.0 ** 13):
                e = int(np.clip(f["e"] + 10 - int(np.floor(np.log2(scaled))), -100, 100))
                self.f16x2_bwd_moves.append((op.layer.name, f["e"], e))
                self._set_bwd_exponent(op, f, entry, e, stream)
        self._judge_direct(direct, stream)
        self.f16x2_checks += 1
        if out:
            self.f16x2_demoted += out
            import warnings
            warnings.warn("training forward (fp16 two-way split): %d layer(s) converted operands outside [%g, %g] and run their forward on the exact bf16 split "
                          "from now on: %s" % (len(out), lo, hi, "; ".join(out)))

    def loss_and_grad(self, labels_ce: torch.Tensor, labels_fg: torch.Tensor, keypoints_yx: torch.Tensor, mask_w=1.0, vertex_w=1.0, proxy_w=1.0,
                      filter_with_segmentation=True, kp: int = 9, filter_high_proxy_errors: bool = False) -> torch.Tensor:
        """Losses of compute_loss on the last forward's output and d loss / d output into self.dout. Returns fp64 [mask, vertex, proxy]."""
        lib = _lib.load()
        B, h, w = self.batch, self.h, self.w
        stream = torch.cuda.current_stream(self.out.device).cuda_stream
        assert labels_ce.dtype == torch.uint8 and labels_fg.dtype == torch.uint8 and keypoints_yx.dtype == torch.float32
        # the power of two on the loss (train_bwd_f16x2): on the GRADIENT only -- the kernels' loss sums do not contain the weights
        self._dout_scale = S = 2.0 ** self.loss_exp
        mask_w, vertex_w, proxy_w = mask_w * S, vertex_w * S, proxy_w * S
        assert tuple(keypoints_yx.shape) == (B, self.seg_dim - 1, kp, 2) and keypoints_yx.is_contiguous()
        oc = self.seg_dim - 1
        if self.pvnet and oc > 1 and self.ver_dim == oc * 2 * kp:   # separated vector fields: per-object slices (compute_loss, train_casapose.py:57,97-125)
            if filter_high_proxy_errors:
                raise NotImplementedError("filter_high_proxy_errors with separated vector fields is not built")
            check(lib.cp_pose_loss_sep_f32(self.out.data_ptr(), self.out_ld, self.seg_dim, kp, labels_ce.data_ptr(), labels_fg.data_ptr(), keypoints_yx.data_ptr(),
                                           oc, B, h, w, 1 if filter_with_segmentation else 0, mask_w, vertex_w, proxy_w, self.loss_ws.data_ptr(),
                                           self.dout.data_ptr(), self.GRAD_LD, self.VERT_OFF, self.loss_sums.data_ptr(), stream), "cp_pose_loss_sep_f32")
            return self.loss_sums
        if 2 * kp > self.ver_dim:
            raise ValueError("the output holds %d vertex channels, fewer than 2 * %d keypoints" % (self.ver_dim, kp))
        check(lib.cp_pose_loss_f32(self.out.data_ptr(), self.out_ld, self.seg_dim, kp, labels_ce.data_ptr(), labels_fg.data_ptr(), keypoints_yx.data_ptr(),
                                   self.seg_dim - 1, B, h, w, 1 if filter_with_segmentation else 0, 1 if filter_high_proxy_errors else 0, mask_w, vertex_w,
                                   proxy_w, self.loss_ws.data_ptr(), self.dout.data_ptr(), self.GRAD_LD, self.VERT_OFF, self.loss_sums.data_ptr(),
                                   self.object_loss_values.data_ptr(), stream), "cp_pose_loss_f32")
        return self.loss_sums

    def kp_loss_and_grad(self, labels_gt: torch.Tensor, gt_xy: torch.Tensor, affine: torch.Tensor, kp_w: float, max_pixel_error: float = 25.0,
                         min_num: int = 50, confidence_regularization: bool = False, vote_with_gt: bool = True, kp: int = 9,
                         min_num_gt: Optional[int] = None, filter_with_gt: bool = True, coords: Optional[torch.Tensor] = None,
                         backward: bool = True, host_loss: Optional[Callable] = None) -> torch.Tensor:
        """keypoint_reprojection_loss (loss_functions.py:207-344, use_bpnp_reprojection_loss=False) on the last forward's
        output; ADDS kp_w * d loss / d output to self.dout (call after loss_and_grad).  gt_xy [B,oc,kp,2]: projected
        ground-truth keypoints in image pixels; affine [B,6]: crop->image map (crop_to_image_affine).  Returns the
        fp64 loss value (device scalar).  `coords` (optional, [B,oc,kp,2] (y,x)) replaces the internal vote (evaluation with the
        component-filtered voter; implies backward=False); min_num_gt / filter_with_gt as in loss_functions.py:221-222,246-252.
        `host_loss(coords, avail) -> (loss, g_yx)` replaces the reprojection kernel (the BPnP variant, whose PnP solve and
        implicit gradient run on the host like the reference's BPNP_fast)."""
        lib = _lib.load()
        B, h, w, K = self.batch, self.h, self.w, self.seg_dim
        oc = K - 1
        out = self.out
        stream = torch.cuda.current_stream(out.device).cuda_stream
        conf_off = K + 2 * kp
        check(lib.cp_argmax_labels(out.data_ptr(), self.out_ld, K, B * h * w, self.est_labels.data_ptr(), stream), "cp_argmax_labels")
        check(lib.cp_kp_stats_f32(out.data_ptr(), self.out_ld, conf_off, labels_gt.data_ptr(), self.est_labels.data_ptr(), B, h, w, K, kp,
                                  self.kp_counts.data_ptr(), self.kp_conf_sums.data_ptr(), stream), "cp_kp_stats_f32")
        vote_labels = labels_gt if vote_with_gt else self.est_labels
        if coords is None:
            check(lib.cp_ls_vote_f32(out.data_ptr(), self.out_ld, 0, K, conf_off, vote_labels.data_ptr(), B, h, w, oc, kp, self.ls_sums.data_ptr(),
                                     self.ls_coords.data_ptr(), stream), "cp_ls_vote_f32")
        else:
            self.ls_coords.copy_(coords.reshape(B, oc, kp, 2))
            backward = False
        # objects_available (loss_functions.py:236-252): > min_num pixels in the estimated AND (filter_with_gt) the ground-truth mask
        avail = self.kp_counts[1, :, 1:] > min_num
        if filter_with_gt:
            avail = avail & (self.kp_counts[0, :, 1:] > (min_num if min_num_gt is None or min_num_gt < 0 else min_num_gt))
        avail = avail.to(torch.float32).contiguous()
        self.objects_available = avail
        if host_loss is not None:
            lv, g = host_loss(self.ls_coords, avail)
            self.ls_g.copy_(torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)).reshape(B, oc, kp, 2))
            self.kp_loss_val.fill_(float(lv))
        else:
            check(lib.cp_kp_reproj_loss_f32(self.ls_coords.data_ptr(), gt_xy.data_ptr(), affine.data_ptr(), avail.data_ptr(), B, oc, kp, max_pixel_error,
                                            kp_w, self.ls_g.data_ptr(), self.kp_loss_val.data_ptr(), stream), "cp_kp_reproj_loss_f32")
        loss = self.kp_loss_val[0]
        if self._dout_scale != 1.0:   # this term's gradient joins one that carries the loss factor
            self.ls_g.mul_(self._dout_scale)
        coef = None
        if confidence_regularization:
            cnt = self.kp_counts[0, :, 1:].sum(dim=1, keepdim=True).double()       # foreground pixels of the target mask
            safe = torch.where(cnt > 0, cnt, torch.ones_like(cnt))
            cl = torch.where(cnt > 0, self.kp_conf_sums / safe, torch.zeros_like(self.kp_conf_sums))
            loss = loss + torch.abs(cl - 0.7).mean()
            coef = (kp_w * self._dout_scale * torch.sign(cl - 0.7) / (B * kp) / safe * (cnt > 0)).to(torch.float32).contiguous()
        if backward:
            check(lib.cp_ls_vote_bwd_f32(out.data_ptr(), self.out_ld, K, conf_off, vote_labels.data_ptr(), B, h, w, oc, kp, self.ls_sums.data_ptr(),
                                         self.ls_g.data_ptr(), self.ls_pu.data_ptr(), labels_gt.data_ptr() if coef is not None else None, _ptr(coef),
                                         self.dout.data_ptr(), self.GRAD_LD, self.VERT_OFF, self.VERT_OFF + 2 * kp, 1, stream), "cp_ls_vote_bwd_f32")
        self._keep_coef = coef
        return loss

    def _gradient_buckets(self):
        """[(first op index, start, end)] of the contiguous slices of the flat gradient that are final once the backward has executed
        the op `first op index` (ops run in reverse): decoder 2 | decoder 1 | stage 4 | the rest of the encoder."""
        st = self.store
        rank = {n: i for i, n in enumerate(forward_order())}
        starts = sorted(rank[b] for b in BUCKET_STARTS if b in rank)

        def bucket_of(layer):
            r = rank.get(layer, len(rank))
            return max([i for i, s0 in enumerate(starts) if r >= s0] + [0])

        nb = len(starts)
        lo, hi, first = [st.size] * nb, [0] * nb, [len(self.ops)] * nb
        for name, (off, shape) in st.offsets.items():
            b = bucket_of(name.split(".")[0])
            n = int(np.prod(shape))
            lo[b], hi[b] = min(lo[b], off), max(hi[b], off + n + ((-n) % 4))
        for i, op in enumerate(self.ops):
            keys = [op.layer.key] if isinstance(op, ConvOp) else ([k for k in (op.gamma_key, op.beta_key) if k] if isinstance(op, BnActOp) else [])
            for k in keys:
                b = bucket_of(k.split(".")[0])
                first[b] = min(first[b], i)
        out = [(first[b], lo[b], hi[b]) for b in range(nb) if hi[b] > lo[b]]
        covered = sorted((a, e) for _, a, e in out)
        assert covered[0][0] == 0 and covered[-1][1] == st.size and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1)), \
            "gradient buckets must tile the flat buffer"
        return out

    def backward(self):
        """Back-propagate self.dout through the tape into store.grad.  With replicas, the SUM all-reduce of each gradient bucket is
        launched (asynchronously, on the collective library's stream) as soon as the backward has passed the bucket's first layer, so
        the exchange of the decoders' gradients overlaps the encoder's backward convolutions; all_reduce_grads() waits for them."""
        stream = torch.cuda.current_stream(self.out.device).cuda_stream
        for t in self.tensors:
            t.has_grad = False
        self._pending = []
        multi = self.group is not None and (self.world_size > 1 or parallel.force_collectives())
        log = self.comm_log
        if (multi or log is not None) and self._buckets is None:
            self._buckets = self._gradient_buckets()
        side = self._side
        main = torch.cuda.current_stream(self.out.device)
        if side is not None:
            side.wait_stream(main)   # the forward's activations and the loss gradient are ready
        # the direct data gradients' operand range: max |dY| by a reduction pass of its own, on the step before a reading of the slots only
        check_now = self._f16x2_mon is not None and (not self._bwd_calibrated or self._f16x2_steps >= F16X2_TRAIN_CHECK_EVERY - 1)
        for i in range(len(self.ops) - 1, -1, -1):
            op = self.ops[i]
            if check_now and isinstance(op, ConvOp) and getattr(op, "bw16", None) is not None and op.bw16["mon"] and not op.bw16["dead"]:
                dy_, ld_ = op._dy()
                check(_lib.load().cp_amax_f32(dy_, op.batch * op.out_h * op.out_w, ld_, op.layer.cout, op.bw16["mon"], stream), "cp_amax_f32(dY %s)" % op.layer.name)
            if side is not None and isinstance(op, ConvOp):
                ev = torch.cuda.Event()
                ev.record(main)                 # dY of this layer is complete on the main stream
                side.wait_event(ev)
                op.backward(stream, wgrad_stream=side.cuda_stream)
            else:
                op.backward(stream)
            if log is not None:
                log.append(("op", i))
            if multi or log is not None:
                for first, a, e in self._buckets:
                    if first == i and not (a == 0):  # the bucket holding bn_data.beta is completed below
                        if log is not None:
                            log.append(("grad_bucket", 4 * (e - a), "async", stream, first))
                        if not multi:
                            continue
                        if side is not None:
                            main.wait_stream(side)   # the bucket's weight gradients come from the side stream
                        self._unscale_grads(stream, a, e)   # (the loss factor is this replica's own: out before the sum over replicas)
                        self._pending.append(parallel.all_reduce_sum_async(self.store.grad[a:e], self.group))
        if side is not None:
            main.wait_stream(side)
        if not self._bwd_calibrated:
            self._calibrate_bwd(stream)
        # d beta of bn_data from the padding-channel entries of conv0's weight gradient (see __init__)
        G = self.conv0.dwp[self.g_idx.reshape(-1)].view(49, 64)            # [tap][cout]
        W0 = self.store.view("conv0.kernel").reshape(49, 3, 64)               # [tap][c][cout]
        dbeta = torch.einsum("tco,to->c", W0.double(), G.double())
        self.store.grad_view("bn_data.beta").copy_(dbeta)
        if multi or log is not None:
            for first, a, e in self._buckets:
                if a == 0:
                    if log is not None:
                        log.append(("grad_bucket", 4 * (e - a), "async", stream, first))
                    if multi:
                        self._unscale_grads(stream, a, e)
                        self._pending.append(parallel.all_reduce_sum_async(self.store.grad[a:e], self.group))
        # the loss carried a power of two (loss_exp): it leaves the flat gradient here -- bucket by bucket in front of each exchange above (the
        # buckets tile the buffer; every replica has its OWN factor, so it must be gone before replicas are summed), in one piece otherwise
        if not multi:
            self._unscale_grads(stream, 0, self.store.grad.numel())
        self._dout_scale = 1.0

    def _unscale_grads(self, stream: int, a: int, e: int):
        if self._dout_scale != 1.0 and e > a:
            g = self.store.grad[a:e]
            check(_lib.load().cp_axpby_f32(g.data_ptr(), 1.0 / self._dout_scale, g.data_ptr(), 0.0, e - a, g.data_ptr(), stream), "cp_axpby_f32(gradient / loss factor)")

    def all_reduce_grads(self):
        """Complete the gradient exchange started by backward() (or run it as one all-reduce if none is pending)."""
        timed = self.comm_timing is not None and self.group is not None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.comm_log is not None:
            self.comm_log.append(("grad_wait", len(getattr(self, "_pending", None) or [])))
        if getattr(self, "_pending", None):
            for h in self._pending:
                h.wait()
            self._pending = []
        else:
            parallel.all_reduce_sum_(self.store.grad, self.group, self.world_size)
        if timed:
            e1.record()
            self.comm_timing["grad_wait"].append((e0, e1))
            self.comm_timing["steps"] += 1

    def train_step(self, img, labels_ce, labels_fg, keypoints_yx, lr: float, cond_labels=None, weights=(1.0, 1.0, 1.0),
                   filter_with_segmentation=True, kp_args: Optional[dict] = None):
        """One optimisation step.  kp_args (optional): keyword arguments of kp_loss_and_grad.  Returns the fp64 device
        vector [mask, vertex, proxy] (and, with kp_args, the keypoint loss as a second value)."""
        stream = torch.cuda.current_stream(img.device).cuda_stream
        if self.comm_log is not None:
            self.comm_log.append(("step_begin",))
        self.forward(img, cond_labels)
        sums = self.loss_and_grad(labels_ce, labels_fg, keypoints_yx, *weights, filter_with_segmentation=filter_with_segmentation)
        kp_loss = self.kp_loss_and_grad(**kp_args) if kp_args is not None else None
        self.backward()
        self.all_reduce_grads()
        self.store.adam_step(lr, stream)
        self.refresh_weights(stream)
        return sums if kp_loss is None else (sums, kp_loss)


def crop_to_image_affine(offsets: np.ndarray) -> np.ndarray:
    """[B,6] row-major 2x3 matrices mapping crop pixels (x,y) back to the original image, the composition applied by
    transform_points_back_tf_batch (ransac_voting.py:124-158) with the offsets layout used at its call site
    (loss_functions.py:276-285): [h_crop, w_crop, -, -, dx, dy, angle_deg, scale, sx, sy]."""
    o = np.asarray(offsets, np.float64)
    hc, wc, dx, dy, ang, sc, sx, sy = o[:, 0], o[:, 1], o[:, 4], o[:, 5], o[:, 6], o[:, 7], o[:, 8], o[:, 9]
    ar = -ang * (np.pi / 180.0)
    a, b = np.cos(ar), np.sin(ar)
    c = (1.0 - a) * sx / 2.0 - b * sy / 2.0
    d = b * sx / 2.0 + (1.0 - a) * sy / 2.0
    A = np.stack([a / sc, b / sc, a * (wc - dx) + b * (hc - dy) + c, -b / sc, a / sc, -b * (wc - dx) + a * (hc - dy) + d], axis=1)
    return A.astype(np.float32)


def project_keypoints(points_3d: np.ndarray, cam: np.ndarray, poses: np.ndarray) -> np.ndarray:
    """project_tf_batch (ransac_voting.py:185-194): points_3d [...,kp,3], cam [3,3], poses [...,3,4] -> [...,kp,2] (x,y)."""
    p = np.asarray(points_3d, np.float64)
    rt = np.asarray(poses, np.float64)
    camp = p @ np.swapaxes(rt[..., :3], -1, -2) + np.swapaxes(rt[..., 3:], -1, -2)
    pix = camp @ np.asarray(cam, np.float64).T
    z = pix[..., 2:]
    return np.where(z != 0, pix[..., :2] / np.where(z != 0, z, 1.0), 0.0).astype(np.float32)
