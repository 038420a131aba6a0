#!/usr/bin/env python3
"""Golden vectors for the host-side geometry of the pose path, produced by the REFERENCE'S OWN NumPy functions executed here (they live in
modules that import TensorFlow / OpenCV at the top, so they are extracted with `ast` and compiled alone -- see make_ndds_frame_golden.py):

    pose_estimation/ransac_voting.py:  get_rotation_matrix_2D (:60-68), transform_points_back (:71-89), project (:161-170)
    utils/geometry_utils.py:           reproject (:7-19), apply_offsets (:22-34), transform_points (:48-57), quaternion_matrix (:144-181),
                                       create_transformation_matrix (:105-113)

    python tests/golden/make_geometry_golden.py        # needs /root/reference; writes tests/golden/geometry_ref.json
"""
import ast
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CASAPOSE_REFERENCE", "/root/reference")


def extract(path, want):
    ns = {"np": np, "math": math}
    tree = ast.parse(open(os.path.join(REF, path)).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    exec(compile(mod, path + " (reference, extracted)", "exec"), ns)
    assert set(want) <= set(ns), sorted(set(want) - set(ns))
    return ns


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference tree at %s" % REF)
    rv = extract("casapose/pose_estimation/ransac_voting.py", ["get_rotation_matrix_2D", "transform_points_back", "project"])
    gu = extract("casapose/utils/geometry_utils.py", ["reproject", "apply_offsets", "get_rotation_matrix_2D", "transform_points", "quaternion_matrix",
                                                      "create_transformation_matrix"])
    rng = np.random.default_rng(11)
    out = {"transform_points_back": [], "project": [], "quaternion_matrix": [], "apply_offsets_round_trip": []}
    for i in range(6):
        # offsets layout of the batch tuple: [h_crop, w_crop, out_h, out_w, dx, dy, angle, scale, sx, sy] (vectorfield_dataset.py:424-435)
        scale = [1.0, 0.75, 1.25, 448 / 420.0, 1.0, 0.9][i]
        off = [float(rng.integers(0, 40)), float(rng.integers(0, 200)), 448.0, 448.0, float(rng.integers(-30, 30)) * (i > 0), float(rng.integers(-30, 30)) * (i > 1),
               float(rng.integers(-25, 25)) * (i > 2), scale, 640.0, 480.0]
        pts = rng.uniform(0, 448, (9, 2))
        # call-site argument order (ransac_voting.py:498-506 map_offsets): w_crop = offsets[1], h_crop = offsets[0], sx = offsets[8], sy = offsets[9]
        back = rv["transform_points_back"](pts.copy(), off[1], off[0], off[8], off[9], off[4], off[5], off[6], off[7])
        out["transform_points_back"].append({"offsets": off, "points_xy": pts.tolist(), "image_xy": np.asarray(back).tolist()})
        # geometry_utils.apply_offsets reads offsets[0] as w_crop and offsets[1] as h_crop: image -> crop with ITS layout; applied to `back` it
        # must return the crop points (the two functions are inverses when fed consistently)
        off_gu = [off[1], off[0]] + off[2:]
        again = gu["apply_offsets"](np.asarray(back, np.float64), off_gu)
        out["apply_offsets_round_trip"].append({"offsets_apply_layout": off_gu, "crop_xy": np.asarray(again).tolist()})
        q = rng.normal(size=4)
        t = rng.normal(size=3) * 100
        out["quaternion_matrix"].append({"q_xyzw": q.tolist(), "t": t.tolist(), "RT": np.asarray(gu["quaternion_matrix"](q, t)).tolist(),
                                         "RT_wxyz_input": np.asarray(gu["quaternion_matrix"](q, t, wxyz_input=True)).tolist(),
                                         "R": np.asarray(gu["quaternion_matrix"](q)).tolist()})
        K = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1.0]])
        RT = np.asarray(gu["quaternion_matrix"](q, [t[0], t[1], 600 + abs(t[2])]))
        xyz = rng.normal(size=(9, 3)) * 40
        xy, cam = rv["project"](xyz, K, RT)
        out["project"].append({"xyz": xyz.tolist(), "K": K.tolist(), "RT": RT.tolist(), "xy": np.asarray(xy).tolist(), "xyz_cam": np.asarray(cam).tolist()})
    M = np.diag([0.1, 0.1, 0.1, 1.0])
    M[:3, 3] = [1, 2, 3]
    p = rng.normal(size=(5, 3))
    out["transform_points"] = {"points": p.tolist(), "M": M.tolist(), "out": np.asarray(gu["transform_points"](p, M)).tolist()}
    out["get_rotation_matrix_2D"] = [{"center": [320.0, 240.0], "angle": a, "M": np.asarray(gu["get_rotation_matrix_2D"]((320.0, 240.0), a)).tolist()} for a in (0, 15, -33)]
    out["loss_weight_handler"] = loss_weight_golden()
    rn = extract("casapose/pose_models/models/resnet.py", ["handle_block_names"])       # resnet.py:20-26: the layer-name stems of the residual units
    # the model registry (pose_models/models_factory.py:9-35): the KEYS and the class each key names, read off the dict literal's syntax tree
    # (its values are attribute references into TensorFlow modules and cannot be evaluated here)
    tree = ast.parse(open(os.path.join(REF, "casapose", "pose_models", "models_factory.py")).read())
    reg = next(n for n in ast.walk(tree) if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "_models")
    out["registry"] = [[k.value, v.attr] for k, v in zip(reg.value.keys, reg.value.values)]
    out["block_names"] = [[s_, b_] + list(rn["handle_block_names"](s_, b_)) for s_ in range(4) for b_ in range(2)]
    json.dump(out, open(os.path.join(HERE, "geometry_ref.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "geometry_ref.json"))




def loss_weight_golden():
    """LossWeightHandler.__init__ / clamp / update (utils/learning_rate_schedules.py:62-109; plain Python, the class statement itself does not
    touch TensorFlow): the per-epoch loss-weight schedule, executed for a few constructor calls and update() sequences."""
    tree = ast.parse(open(os.path.join(REF, "casapose", "utils", "learning_rate_schedules.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LossWeightHandler")
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("__init__", "clamp", "update")]
    ns = {}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "learning_rate_schedules.py (reference, extracted)", "exec"), ns)
    H = ns["LossWeightHandler"]
    cases = [dict(args=[], kw={}, updates=3),
             dict(args=[1.0, 0.5, 0.015, 0.007], kw={}, updates=2),                                                      # config_8.ini's weights, positional
             dict(args=[], kw=dict(mask_loss_weight=2.0, mask_loss_factor=1.5, vertex_loss_weight=8.0, vertex_loss_factor=1.2, proxy_loss_weight=0.02,
                                   proxy_loss_factor=1.3, kp_loss_weight=0.1, kp_loss_factor=0.5, kp_loss_borders=(0.04, 2.5),
                                   filter_vertex_with_segmentation=True), updates=6),
             dict(args=[1.0, 1.0, 0.01, 1.0, 0.9, 0.9, 0.9, 0.9, (0.5, 2.5), (0.7, 10.0), (0.009, 0.025), (0.0, 2.5), True, True], kw={}, updates=4)]
    out = []
    for c in cases:
        h = H(*c["args"], **c["kw"])
        trace = []
        for _ in range(c["updates"] + 1):
            trace.append([h.mask_loss_weight, h.vertex_loss_weight, h.proxy_loss_weight, h.kp_loss_weight])
            h.update()
        out.append({"args": c["args"], "kw": c["kw"], "trace": trace, "flags": [h.filter_vertex_with_segmentation, h.filter_high_proxy_errors]})
    return out


if __name__ == "__main__":
    main()
