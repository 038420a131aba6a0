#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the NumPy oracle.

SELF-GOLDENS: the reference cannot be imported here (TensorFlow 2.9.1 / tfa / cv2 missing,
SURVEY.md F2) and ships no vectors, so these files freeze the oracle's fp64 answers on seeded
inputs -- they protect against drift of the oracle and give the GPU tests fixed targets; they
are NOT outputs of the reference.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import casapose_oracle as O  # noqa: E402


def forward_case():
    k, v, h, w = 5, 27, 32, 48
    p = O.init_params(k, v, seed=1237, dtype=np.float64)
    rng = np.random.default_rng(11)
    img = rng.uniform(-1, 1, (1, h, w, 3)).astype(np.float32)
    lab = np.zeros((1, h, w), np.uint8)
    lab[:, 4:20, 6:30] = 1
    lab[:, 14:30, 22:44] = 2
    lab[:, 2:10, 34:46] = 3
    lab[:, 24:31, 2:12] = 4
    seg = O.onehot_from_labels(lab.astype(np.int64), k)
    out = O.casapose_c_gcu5(p, img.astype(np.float64), seg_input=seg)
    checksum = float(sum(np.abs(a).sum() for a in p.values()))
    np.savez_compressed(os.path.join(HERE, "forward_gcu5_k5_32x48.npz"), image=img, labels=lab, output=out.astype(np.float32),
                        param_seed=1237, param_abs_sum=checksum)


def layer_cases():
    rng = np.random.default_rng(21)
    b, h, w, k, cin, cout = 1, 12, 16, 4, 32, 32
    lab = np.zeros((b, h, w), np.uint8)
    lab[:, 2:9, 3:12] = 1
    lab[:, 6:11, 8:15] = 2
    lab[:, 0:3, 12:16] = 3
    mask = O.onehot_from_labels(lab.astype(np.int64), k)
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cin, 3, 3, cout)) / 17.0).astype(np.float32)
    pc = O.partial_convolution(x.astype(np.float64), wt.astype(np.float64), mask)
    lo = rng.standard_normal((b, h // 2, w // 2, cin)).astype(np.float32)
    gu = O.guided_upsampling(lo.astype(np.float64), O.half_size(mask), mask)
    gb = O.guided_bilinear_upsampling(lo.astype(np.float64), O.half_size(mask), mask)
    bl = O.upsample_bilinear_x2(lo.astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "layers_k4_12x16.npz"), labels=lab, x=x, weights_ihwo=wt, partial_conv=pc.astype(np.float32),
                        low=lo, guided_up=gu.astype(np.float32), guided_bilinear_up=gb.astype(np.float32), bilinear_up=bl.astype(np.float32))


def voting_case():
    b, h, w, objs = 1, 60, 80, 8
    seg, direct, conf, labels, kps = O.synthetic_voting_inputs(b, h, w, num_obj=objs, seed=31)
    ls = O.ls_voting(seg, direct, conf)
    # filtered variant: add a detached speck to object 1 and a second blob to object 2
    seg_f = seg.copy()
    seg_f[0, 0:2, 0:3, :] = 0.0
    seg_f[0, 0:2, 0:3, 1] = 5.0
    ls_f = O.ls_voting(seg_f, direct, conf, filter_estimates=True)
    rng = np.random.default_rng(32)
    hyp = 128
    draws = rng.integers(0, 2**31 - 1, (2, b, objs, hyp, 9, 2), dtype=np.int64).astype(np.int32)
    rs = np.zeros((b, objs, 9, 2), np.float32)
    rounds = np.zeros((b, objs), np.int32)
    for o in range(objs):
        m = (labels[0] == o + 1).astype(np.float32)
        tn = int(m.sum())
        idx = [draws[r, 0, o].astype(np.int64) % max(tn, 1) for r in range(2)]
        rs[0, o], rounds[0, o] = O.ransac_voting_single(m, direct[0].reshape(h, w, 9, 2), idx, max_iter=2)
    np.savez_compressed(os.path.join(HERE, "voting_8obj_60x80.npz"), seed=31, ls_keypoints=ls, ls_keypoints_filtered=ls_f,
                        ransac_draws=draws, ransac_keypoints=rs, ransac_rounds=rounds, true_keypoints=kps.astype(np.float32))


def ellipse_labels(h, w, objects, seed):
    """objects non-overlapping ellipses on a grid, one per class, rest background (SURVEY 8d training inputs)."""
    rng = np.random.default_rng(seed)
    cols = int(np.ceil(np.sqrt(objects * w / h)))
    rows = int(np.ceil(objects / cols))
    yy, xx = np.mgrid[0:h, 0:w]
    lab = np.zeros((h, w), np.uint8)
    for o in range(objects):
        cy, cx = (o // cols + 0.5) * h / rows, (o % cols + 0.5) * w / cols
        ry, rx = rng.uniform(0.25, 0.45) * h / rows, rng.uniform(0.25, 0.45) * w / cols
        lab[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = o + 1
    return lab


def fullsize_case(name, k, h, w, seed, samples=8192):
    """ONE image at a BASELINE.json size through the fp64 oracle (minutes on this machine, once): the GPU tests cannot afford the
    oracle at that size, so the fixture keeps (a) the oracle's own arg-max label map, (b) a sample of complete output records of the
    estimated-mask forward, (c) the same sample of the forward conditioned on a GIVEN synthetic label map (ellipses), (d) the logits'
    top-2 margin at every pixel (uint8-quantised flag: margin < 1e-3 of the range) so that label disagreements can be attributed
    to near-ties.  Inputs are regenerated from the seeds by the test; checksums guard that."""
    v = 27
    p = O.init_params(k, v, seed=seed, dtype=np.float64)
    img = np.random.default_rng(seed).uniform(-1, 1, (1, h, w, 3)).astype(np.float32)
    given = ellipse_labels(h, w, k - 1, seed)[None]
    out_est = O.casapose_c_gcu5(p, img.astype(np.float64))
    out_giv = O.casapose_c_gcu5(p, img.astype(np.float64), seg_input=O.onehot_from_labels(given.astype(np.int64), k))
    logits = out_est[0, ..., :k]
    top2 = np.sort(logits, -1)[..., -2:]
    near_tie = ((top2[..., 1] - top2[..., 0]) < 1e-3 * np.abs(logits).max()).astype(np.uint8)
    rng = np.random.default_rng(seed + 1)
    ys, xs = rng.integers(0, h, samples), rng.integers(0, w, samples)
    np.savez_compressed(os.path.join(HERE, name), seed=seed, classes=k, height=h, width=w,
                        image_abs_sum=float(np.abs(img.astype(np.float64)).sum()), param_abs_sum=float(sum(np.abs(a).sum() for a in p.values())),
                        labels_estimated=logits.argmax(-1).astype(np.uint8), near_tie=np.packbits(near_tie), labels_given=given[0],
                        sample_y=ys.astype(np.int16), sample_x=xs.astype(np.int16),
                        records_estimated=out_est[0, ys, xs].astype(np.float32), records_given=out_giv[0, ys, xs].astype(np.float32),
                        logit_range=float(np.abs(logits).max()), field_range=float(np.abs(out_giv[0, ..., k:]).max()))


if __name__ == "__main__":
    if "--fullsize" in sys.argv:  # minutes per image: run on demand, the small cases below are not touched
        fullsize_case("fullsize_gcu5_k9_480x640.npz", 9, 480, 640, 1237)       # BASELINE configs[1]: 8-object LMO inference size
        fullsize_case("fullsize_gcu5_k14_448x448.npz", 14, 448, 448, 1313)     # configs[4]: 13-object network at the training size
    else:
        forward_case()
        layer_cases()
        voting_case()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
