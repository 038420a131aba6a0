"""ransac_voting_layer_all_masks with the reference's signature
(casapose/pose_estimation/ransac_voting.py:446-484), computed by cp_ransac_vote_f32.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib
from .._lib import check

# The seed of a call made with a DEVICE generator is f(initial_seed, the generator's Philox offset), and the call advances that offset by four -- host-side
# bookkeeping of the generator (get_offset / set_offset never touch the device), so the voter consumes the generator like any other sampler without a
# synchronisation: `manual_seed(s)` restarts the sequence, `get_state()` / `set_state()` capture and restore it (a private call counter
# keyed by id(generator) did neither, and could be inherited by a new generator at a recycled id).  Where a build has no offset accessors the old
# counter remains as a fallback, now tied to the initial_seed it counts for.
_DEVICE_GENERATOR_CALLS: dict = {}   # fallback only: id(generator) -> [initial_seed, calls]


def _device_generator_seed(generator: torch.Generator) -> int:
    seed0 = int(generator.initial_seed())
    try:
        off = int(generator.get_offset())
        generator.set_offset(off + 4)
        return (seed0 * 0x9E3779B97F4A7C15 + off // 4 + 1) % (2**62)
    except (RuntimeError, AttributeError, NotImplementedError):
        pass
    key = id(generator)
    ent = _DEVICE_GENERATOR_CALLS.pop(key, None)
    if ent is None or ent[0] != seed0:
        ent = [seed0, 0]
    _DEVICE_GENERATOR_CALLS[key] = ent            # (re-inserted: most recently used last)
    while len(_DEVICE_GENERATOR_CALLS) > 64:
        _DEVICE_GENERATOR_CALLS.pop(next(iter(_DEVICE_GENERATOR_CALLS)))
    ent[1] += 1
    return (seed0 * 0x9E3779B97F4A7C15 + ent[1]) % (2**62)


def ransac_voting_layer_all_masks(mask: torch.Tensor, vertex: torch.Tensor, round_hyp_num: int, inlier_thresh: float = 0.99,
                                  confidence: float = 0.99, max_iter: int = 20, min_num: int = 5, max_num: int = 30000,
                                  draws: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                                  return_rounds: bool = False):
    """mask [b,h,w,oc] one-hot object masks (no background channel), vertex [b,h,w,vn,2] or
    [b,h,w,vn*2] in (dy,dx) order.  Returns [b,oc,vn,2] keypoints in (x,y).

    By default (round 4) every random number is made inside the library from one 64-bit seed (cp_ransac_vote_seeded_f32: counter-based draws per round,
    and the random thinning of objects above `max_num` pixels, :295-301, inside the compaction; the `min_num` gate of :290-292 is applied to the
    UN-thinned count, as there) -- the seed is drawn on the HOST: from `generator` when it is a CPU generator, from torch's default CPU generator
    when none is given, and for a CUDA generator from its initial_seed() and a per-generator call counter (drawing from a device generator would
    synchronise the stream: ADVICE round 4); no draw tensor exists.  The random stream of a given seed differs from round 3's torch.randint
    draws (INTEGRATION.md, section on the voters).  `draws` (int32 [max_iter,b,oc,round_hyp_num,vn,2],
    values in [0,2^31)) replaces the pixel-pair draws of :319-321 -- tests inject them; on that path the thinning is done here with torch (it needs
    the counts on the host)."""
    if not mask.is_cuda:
        raise _lib.CasaposeHipError("ransac_voting_layer_all_masks needs CUDA (ROCm) tensors; there is no CPU fallback")
    lib = _lib.load()
    b, h, w, oc = mask.shape
    vert = vertex.reshape(b, h, w, -1).to(torch.float32).contiguous()
    vn = vert.shape[3] // 2
    stream = torch.cuda.current_stream(mask.device).cuda_stream
    labels = torch.empty(b, h, w, dtype=torch.uint8, device=mask.device)
    counts = torch.empty(b, oc, dtype=torch.int32, device=mask.device)
    check(lib.cp_mask_to_labels_f32(mask.to(torch.float32).contiguous().data_ptr(), b, h, w, oc, labels.data_ptr(), counts.data_ptr(), stream),
          "cp_mask_to_labels_f32")   # one pass instead of five PyTorch reductions / elementwise kernels over [b,h,w,oc]
    ws = torch.empty(lib.cp_ransac_workspace_bytes(b, h, w, oc, vn, round_hyp_num), dtype=torch.uint8, device=mask.device)
    out = torch.empty(b, oc, vn, 2, dtype=torch.float32, device=mask.device)
    rounds = torch.empty(b, oc, dtype=torch.int32, device=mask.device)
    if draws is None:
        if generator is not None and generator.device.type != "cpu":
            seed = _device_generator_seed(generator)
        else:
            seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64, generator=generator).item())   # CPU generator: no device synchronisation
        check(lib.cp_ransac_vote_seeded_f32(labels.data_ptr(), vert.data_ptr(), vert.shape[3], 0, b, h, w, oc, vn, seed, round_hyp_num, float(inlier_thresh),
                                            float(confidence), int(max_iter), int(min_num), int(max_num), ws.data_ptr(), out.data_ptr(), rounds.data_ptr(), stream),
              "cp_ransac_vote_seeded_f32")
        return (out, rounds) if return_rounds else out
    if int(counts.max()) > max_num:
        # random down-sampling of large masks (:295-301): keep a pixel with probability max_num / count
        keep_p = (max_num / counts.clamp(min=1).to(torch.float32)).clamp(max=1.0)  # [b,oc]
        u = torch.rand(b, h, w, device=mask.device, generator=generator)
        lab_l = labels.long()
        p_pix = torch.cat([torch.ones(b, 1, device=mask.device), keep_p], dim=1).gather(1, lab_l.reshape(b, -1)).reshape(b, h, w)
        labels = torch.where(u < p_pix, labels, torch.zeros_like(labels)).contiguous()
    if tuple(draws.shape) != (max_iter, b, oc, round_hyp_num, vn, 2) or draws.dtype != torch.int32:
        raise ValueError("draws must be int32 with shape [max_iter,b,oc,hyp,vn,2]")
    draws = draws.contiguous()
    check(lib.cp_ransac_vote_f32(labels.data_ptr(), vert.data_ptr(), vert.shape[3], 0, b, h, w, oc, vn, draws.data_ptr(), round_hyp_num,
                                 float(inlier_thresh), float(confidence), int(max_iter), int(min_num), int(max_num), ws.data_ptr(),
                                 out.data_ptr(), rounds.data_ptr(), stream), "cp_ransac_vote_f32")
    return (out, rounds) if return_rounds else out
