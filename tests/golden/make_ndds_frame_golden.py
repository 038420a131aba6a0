#!/usr/bin/env python3
"""Golden vectors for the per-frame NDDS reader produced by the REFERENCE'S OWN CODE, executed here.

`casapose/data_handler/vectorfield_dataset.py` imports TensorFlow, tensorflow-addons and trimesh at its top and cannot be imported in this
container -- but the functions that define the on-disk format and the batch tuple are plain NumPy / json:

    VectorfieldDataset.load_json_minimal   (:545-597)   per-frame <name>.json -> keypoints, poses, classes, pixel counts, visibility filter
    VectorfieldDataset.load_json_classes   (:599-615)   _object_settings.json -> segmentation ids, fixed model transforms
    VectorfieldDataset.load_json_camera    (:617-631)   _camera_settings.json -> K
    VectorfieldDataset.apply_preprocessing (:291-509)   crop / scale geometry, keypoint reprojection, pose matrices, offsets, affine, image id
    geometry_utils.{reproject, get_rotation_matrix_2D, transform_points, quaternion_matrix}   (utils/geometry_utils.py:7-58,144-181)

This script parses the two reference files with `ast`, compiles ONLY those function definitions (nothing of the reference's text is written
anywhere) and runs them on a hand-written frame -- key names and nesting are therefore exactly what the reference READS, not what this
repository's `write_ndds_scene()` writes.  Outputs go to tests/golden/ndds_frame_ref.json; the frame itself (json files + two small PNGs) to
tests/golden/ndds_frame/.  tests/test_ndds_reader.py runs OUR reader on the same frame and compares field by field.

Not executed (TensorFlow): `set_new_labels` (a 10-line list of [segmentation id, index + 1] pairs, restated in NumPy below and marked as such in
the output), `load_images` / `image_transformation` (tf.image decoding, tfa warps).  Meshes: the keypoints are the reference's LM files
(tests/golden/ref_data); `volume` (trimesh's bounding-box corner ORDER) is not pinned -- the same eight corners are handed to both sides.

    python tests/golden/make_ndds_frame_golden.py            # needs /root/reference (this container only)
"""
import ast
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CASAPOSE_REFERENCE", "/root/reference")
FRAME_ROOT = os.path.join(HERE, "ndds_frame")
SCENE = os.path.join(FRAME_ROOT, "data", "lmo_test", "000002")
NAMES = ["obj_000001", "obj_000005", "obj_000009", "obj_000012"]      # objects of interest (obj_000012 is absent from the frame)
H, W = 480, 640
K = dict(fx=572.4114, fy=573.57043, cx=325.2611, cy=242.04899)         # the LM camera


def reference_functions():
    """compile the TF-free functions of the reference into a namespace (np / math / json / os only)"""
    ns = {"np": np, "math": math, "json": json, "os": os}
    tree = ast.parse(open(os.path.join(REF, "casapose", "utils", "geometry_utils.py")).read())
    want = {"reproject", "get_rotation_matrix_2D", "transform_points", "quaternion_matrix"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    exec(compile(mod, "geometry_utils.py (reference, extracted)", "exec"), ns)
    assert want <= set(ns)
    tree = ast.parse(open(os.path.join(REF, "casapose", "data_handler", "vectorfield_dataset.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VectorfieldDataset")
    want = {"load_json_minimal", "load_json_classes", "load_json_camera", "apply_preprocessing", "load_image_data"}
    mod = ast.Module(body=[n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    # the one compatibility edit: the reference targets NumPy 1.x, where the dtype alias "unicode_" exists (:485); NumPy 2 spells it "str_"
    for node in ast.walk(mod):
        if isinstance(node, ast.Constant) and node.value == "unicode_":
            node.value = "str_"
    exec(compile(mod, "vectorfield_dataset.py (reference, extracted)", "exec"), ns)
    assert want <= set(ns)
    # file discovery and the train / validation split file: utils/dataset_utils.py load_split (:462-475), write_json_split (:478-493) and the
    # JSON writer they use, utils/io_utils.py to_json (:9-51) -- plus the names load_image_data takes from its module's imports
    import glob
    from itertools import compress
    from os.path import exists

    ns.update(glob=glob, compress=compress, exists=exists)
    for path, names in (("casapose/utils/io_utils.py", {"to_json"}), ("casapose/utils/dataset_utils.py", {"load_split", "write_json_split"})):
        tree = ast.parse(open(os.path.join(REF, path)).read())
        mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names], type_ignores=[])
        exec(compile(mod, path + " (reference, extracted)", "exec"), ns)
        assert names <= set(ns)
    return ns


def read_ply_ascii(path):
    txt = open(path).read().split("end_header")[1].strip().splitlines()
    return np.array([[float(v) for v in l.split()[:3]] for l in txt], np.float64)


def bbox_corners(v):
    lo, hi = v.min(0), v.max(0)
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float64)


def quat_xyzw(axis, deg):
    a = np.asarray(axis, float)
    a /= np.linalg.norm(a)
    s = math.sin(math.radians(deg) / 2)
    return [float(a[0] * s), float(a[1] * s), float(a[2] * s), float(math.cos(math.radians(deg) / 2))]


def project(Kd, q, t, pts, ns):
    RT = ns["quaternion_matrix"](q, t)
    cam = (RT[:, :3] @ pts.T).T + RT[:, 3]
    return np.stack([Kd["fx"] * cam[:, 0] / cam[:, 2] + Kd["cx"], Kd["fy"] * cam[:, 1] / cam[:, 2] + Kd["cy"]], 1)


def write_frame(ns):
    """the hand-written frame: every key below is one the reference's reader dereferences (load_json_minimal :553-588, load_json_classes
    :607-613, load_json_camera :623-629); values are chosen to exercise its branches"""
    from PIL import Image

    os.makedirs(SCENE, exist_ok=True)
    kp_dir = os.path.join(HERE, "ref_data", "lm_models_eval")
    kps = {n: read_ply_ascii(os.path.join(kp_dir, n + "_keypoints.ply")) for n in NAMES}
    # fixed_model_transform is stored TRANSPOSED in the file (the reader transposes it back, :611-613); obj_000005 carries a scale of 0.1 (mm -> cm)
    fixed = {"obj_000001": np.eye(4), "obj_000005": np.diag([0.1, 0.1, 0.1, 1.0]), "obj_000009": np.eye(4), "obj_000012": np.eye(4),
             "obj_000002": np.eye(4)}
    fixed["obj_000009"][:3, 3] = [1.0, -2.0, 0.5]
    seg_ids = {"obj_000001": 21, "obj_000005": 106, "obj_000009": 191, "obj_000012": 255, "obj_000002": 42}
    json.dump({"exported_object_classes": list(seg_ids),
               "exported_objects": [{"class": n, "segmentation_class_id": seg_ids[n], "fixed_model_transform": fixed[n].T.tolist()} for n in seg_ids]},
              open(os.path.join(SCENE, "_object_settings.json"), "w"), indent=1)
    json.dump({"camera_settings": [{"name": "camera", "intrinsic_settings": dict(K, s=0), "captured_image_size": {"width": W, "height": H}}]},
              open(os.path.join(SCENE, "_camera_settings.json"), "w"), indent=1)
    # instances: (class, quaternion xyzw, location, visibility, px_count_all or None).  Locations are in the units of the TRANSFORMED model.
    inst = [("obj_000001", quat_xyzw([0.2, 1, 0.1], 35), [-60.0, 20.0, 800.0], 0.93, 4211),
            ("obj_000005", quat_xyzw([1, 0, 0], 80), [3.0, -4.0, 70.0], 0.04, 12),        # first instance of the class: below the 0.10 visibility cut
            ("obj_000005", quat_xyzw([0, 1, 1], 140), [9.0, 6.0, 95.0], 0.71, 2950),       # second instance: the one the reader must keep
            ("obj_000009", quat_xyzw([1, 1, 0], 10), [-400.0, -150.0, 900.0], 0.55, None), # projects left of the 448x448 centre crop; no px_count_all key
            ("obj_000002", quat_xyzw([0, 0, 1], 5), [250.0, -100.0, 1000.0], 1.0, 777)]     # a class that is not an object of interest
    objects = []
    seg = np.zeros((H, W), np.uint8)
    img = np.full((H, W, 3), 90, np.uint8)
    img[::16, :, :] = 130
    for cls, q, t, vis, px in inst:
        pts3 = ns["transform_points"](kps.get(cls, kps["obj_000001"]), fixed[cls])
        uv = project(K, q, t, np.asarray(pts3), ns)
        o = {"class": cls, "visibility": vis, "location": t, "quaternion_xyzw": q, "keypoints_2d": uv.tolist(), "keypoints_3d": np.asarray(pts3).tolist(),
             "instance_id": len(objects)}
        if px is not None:
            o["px_count_all"] = px
        objects.append(o)
        if vis > 0.1:   # a blob of the class's segmentation id around the projected centre
            cx, cy = int(round(uv[0, 0])), int(round(uv[0, 1]))
            y0, y1, x0, x1 = max(cy - 30, 0), min(cy + 30, H), max(cx - 40, 0), min(cx + 40, W)
            seg[y0:y1, x0:x1] = seg_ids[cls]
            img[y0:y1, x0:x1] = (seg_ids[cls], 255 - seg_ids[cls], 128)
    json.dump({"camera_data": {"location_worldframe": [0, 0, 0], "quaternion_xyzw_worldframe": [0, 0, 0, 1]}, "objects": objects},
              open(os.path.join(SCENE, "000017.json"), "w"), indent=1)
    Image.fromarray(img).save(os.path.join(SCENE, "000017.png"), optimize=True)
    Image.fromarray(seg).save(os.path.join(SCENE, "000017.seg.png"), optimize=True)
    return kps


def discovery_golden(ns, Ref):
    """load_image_data (:682-746) on a small tree under tests/golden/ndds_frame/discovery/: two leaf folders; images as .png, .jpg (the reader
    falls back png -> bmp -> jpg), a seg.png without its .json (skipped), an image without a seg.png (never seen).  Then the same with the
    TRAIN and the VALIDATION half of a split file that the reference's write_json_split / to_json wrote (copied into the fixture as
    split_settings_written_by_reference.json: the test installs it as _split_settings.json in a temporary copy of the tree)."""
    import shutil
    import tempfile

    from PIL import Image

    root = os.path.join(FRAME_ROOT, "discovery")
    shutil.rmtree(root, ignore_errors=True)
    tiny = Image.fromarray(np.zeros((4, 6, 3), np.uint8))
    seg = Image.fromarray(np.zeros((4, 6), np.uint8))
    leaves = {"sceneA/000001": ["000000.png", "000001.png", "000002.jpg", "000003.png", "000004.png"], "sceneB": ["000010.jpg", "000011.png"]}
    for leaf, files in leaves.items():
        d = os.path.join(root, leaf)
        os.makedirs(d)
        shutil.copy(os.path.join(SCENE, "_object_settings.json"), d)
        shutil.copy(os.path.join(SCENE, "_camera_settings.json"), d)
        for f in files:
            stem = f.split(".")[0]
            tiny.save(os.path.join(d, f))
            seg.save(os.path.join(d, stem + ".seg.png"))
            json.dump({"objects": []}, open(os.path.join(d, stem + ".json"), "w"))
    d = os.path.join(root, "sceneA", "000001")
    seg.save(os.path.join(d, "000005.seg.png"))                     # no image, no json: skipped
    tiny.save(os.path.join(d, "000006.png"))                        # no seg.png: never listed
    seg.save(os.path.join(d, "000007.seg.png"))
    tiny.save(os.path.join(d, "000007.png"))                        # image + seg but no json: skipped
    gold = {"root": os.path.relpath(root, HERE)}

    def run(tree, **flags):
        r = Ref()
        r.use_train_split, r.use_validation_split, r.train_validation_split = flags.get("train", False), flags.get("val", False), 0.6
        imgs = []
        for name in sorted(os.listdir(tree)):                         # load_data (:95-106) walks os.listdir(root); order is the file system's
            imgs += r.load_image_data(tree + "/" + name)[0]
        return sorted([os.path.relpath(i[0], tree), i[1], os.path.relpath(i[2], tree), os.path.relpath(i[3], tree), os.path.relpath(i[4], tree)] for i in imgs)

    gold["all"] = run(root)
    with tempfile.TemporaryDirectory() as tmp:
        tree = os.path.join(tmp, "d")
        shutil.copytree(root, tree)
        np.random.seed(7)
        gold["train"] = run(tree, train=True)                          # writes _split_settings.json in every leaf (write_json_split)
        gold["val"] = run(tree, val=True)                              # re-reads them (load_split)
        for leaf in leaves:
            shutil.copy(os.path.join(tree, leaf, "_split_settings.json"), os.path.join(root, leaf, "split_settings_written_by_reference.json"))
    return gold


class Reader:
    """the state apply_preprocessing dereferences on `self` (vectorfield_dataset.py:60-130), filled by the reference's own loaders"""


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference tree at %s" % REF)
    ns = reference_functions()
    kps = write_frame(ns)
    Ref = type("RefReader", (Reader,), {n: ns[n] for n in ("load_json_minimal", "load_json_classes", "load_json_camera", "apply_preprocessing", "load_image_data")})
    info = json.load(open(os.path.join(HERE, "ref_data", "lm_models_eval", "models_info.json")))
    rng = np.random.default_rng(0)
    out = {"frame": os.path.relpath(SCENE, HERE), "names": NAMES, "cases": [],
           "not_from_reference_code": ["new_labels (set_new_labels uses tf.constant: restated in NumPy)", "meshes.volume corner order (trimesh)"]}
    for vis_filter, imagesize, crop in ((True, (480, 640), 1.0), (True, (448, 448), 0.933333333), (False, (448, 448), 0.933333333)):
        r = Ref()
        r.objectsofinterest = NAMES
        r.visibility_filter, r.wxyz_quaterion_input = vis_filter, False
        r.random_crop, r.random_translation, r.random_rotation = False, (0.0, 0.0), 0.0
        r.meshes = {}
        for n in NAMES:
            b = info[n]
            lo = np.array([b["min_x"], b["min_y"], b["min_z"]])
            r.meshes[n] = {"keypoints": kps[n], "volume": bbox_corners(np.stack([lo, lo + [b["size_x"], b["size_y"], b["size_z"]]])), "diameter": b["diameter"]}
        labels, fixed = r.load_json_classes(os.path.join(SCENE, "_object_settings.json"))
        r.class_labels, r.fixed_transformations = {SCENE: labels}, {SCENE: fixed}
        r.camera_data = {SCENE: r.load_json_camera(os.path.join(SCENE, "_camera_settings.json"))}
        r.set_new_labels = lambda seg_img, object_labels: np.array([[l, i + 1] if l is not None else [0, 0] for i, l in enumerate(object_labels)])
        case = {"visibility_filter": vis_filter, "imagesize": list(imagesize), "cropratio": crop}
        try:
            np.random.seed(3)
            res = r.apply_preprocessing(np.zeros((H, W, 3), np.uint8), b"000017.png", os.path.join(SCENE, "000017.json"), np.zeros((H, W, 1), np.uint8),
                                        SCENE.encode(), imagesize, crop, 1, 9)
            keys = ("img", "seg_img", "keypoints2d", "keypoints3d", "camera_data", "diameters", "offsets", "affine", "cuboid3d", "transform_mats",
                    "pixel_gt_count", "image_id", "new_labels")
            for k, v in zip(keys, res):
                if k in ("img", "seg_img"):
                    continue
                case[k] = np.asarray(v).tolist()
        except Exception as exc:   # two VISIBLE instances of one class with max_instance_count = 1: the reference's own lists are ragged (:392-413, :478)
            case["raises"] = "%s: %s" % (type(exc).__name__, str(exc)[:120])
        out["cases"].append(case)
        minimal = r.load_json_minimal(os.path.join(SCENE, "000017.json"))
        case["load_json_minimal"] = {"objectClasses": minimal["objectClasses"], "px_count_all": minimal["px_count_all"],
                                     "poses_loc": np.asarray(minimal["poses_loc"]).tolist(), "n_keypoints2d": [len(k) for k in minimal["keypoints2d"]]}
    out["discovery"] = discovery_golden(ns, Ref)
    out["load_json_classes"] = {"labels": labels, "fixed": {k: np.asarray(v).tolist() for k, v in fixed.items()}}
    out["load_json_camera"] = np.asarray(r.camera_data[SCENE]).tolist()
    json.dump(out, open(os.path.join(HERE, "ndds_frame_ref.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "ndds_frame_ref.json"), "and the frame under", FRAME_ROOT)
    for c in out["cases"]:
        print(c["visibility_filter"], c["imagesize"], "raises" in c and c["raises"] or "ok")
    del rng


if __name__ == "__main__":
    main()
