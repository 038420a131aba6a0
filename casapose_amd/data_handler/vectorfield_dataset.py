"""NDDS / converted-BOP dataset reader producing the batch tuple of the training and evaluation steps -- the counterpart of
VectorfieldDataset (casapose/data_handler/vectorfield_dataset.py:24-180,291-509,545-762,764-1013,1047-1074) without
TensorFlow: PIL for images, NumPy for geometry, plain Python iteration instead of tf.data.

On-disk layout read (the reference's, :682-746):
    <root>/<scene...>/NNNNNN.png|bmp|jpg   colour image            NNNNNN.seg.png  uint8 segmentation-class ids
    <root>/<scene...>/NNNNNN.json           objects[]: class, visibility, px_count_all, keypoints_2d [[x,y]...], keypoints_3d,
                                            quaternion_xyzw, location
    <root>/<scene...>/_object_settings.json exported_objects[]: class, segmentation_class_id, fixed_model_transform (4x4, stored transposed)
    <root>/<scene...>/_camera_settings.json camera_settings[0].intrinsic_settings {fx, fy, cx, cy}
    <meshes>/<obj>/<obj>.ply|obj , <obj>_keypoints.ply , <meshes>/models_info.json {<obj>: {diameter}}

Per image (apply_preprocessing, :291-509): crop of height round(H*crop_factor) and the output aspect ratio (random or centred),
optional random rotation / translation about the image centre, resize to `imagesize`; 2-D keypoints follow the same map and
come out in (y,x); poses from quaternion_xyzw + location; offsets = [h_crop, w_crop, crop_h, crop_w, dx, dy, angle, scale, W, H];
segmentation ids are re-labelled to 1..len(objectsofinterest) in the order of `objectsofinterest` (:748-762, :996-1008).
Batches are dicts with the field names used by casapose_amd.training (img_batch indices of train_casapose.py:496-507):
img[0] target_seg[1] target_vert[3] keypoints3d[4] cam_mat[5] diameters[6] offsets[7] filtered_seg[8] poses_gt[10]
pixel_gt_count[11] image_id[12].

Not carried over: imgaug pipelines and hue / saturation jitter (data augmentation is outside SURVEY 8); brightness / contrast
jitter and the additive noise of image_augmentation (:259-272) are.  Only one instance per object is read, like the reference
(`max_count = 1`, :795).  Parity with the reference's loader is untested here (no dataset on this machine): the round trip
against write_ndds_scene() below pins the conventions as read from the reference's code.
"""
from __future__ import annotations

import glob
import json
import math
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------
# small geometry / file helpers
# ------------------------------------------------------------------------------------------------
def quaternion_matrix(quaternion_xyzw, translation=None, wxyz_input=False) -> np.ndarray:
    """rotation (3x3) or pose (3x4) from a quaternion (utils/geometry_utils.py:144-181)."""
    q = np.array(quaternion_xyzw, dtype=np.float64)
    if not wxyz_input:
        q = np.array([q[3], q[0], q[1], q[2]])
    n = float(q @ q)
    if n < 1e-4:
        return np.identity(4)
    q = q * math.sqrt(2.0 / n)
    q = np.outer(q, q)
    R = np.array([[1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0]],
                  [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0]],
                  [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2]]])
    if translation is None:
        return R
    return np.concatenate([R, np.asarray(translation, np.float64).reshape(3, 1)], axis=1)


def matrix_to_quaternion_xyzw(R: np.ndarray) -> np.ndarray:
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w, x, y, z = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        v = [0.0, 0.0, 0.0]
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        w = (R[k, j] - R[j, k]) / s
        x, y, z = v
    return np.array([x, y, z, w])


def get_rotation_matrix_2D(center, angle_deg) -> np.ndarray:
    a_, b_ = math.cos(angle_deg * math.pi / 180), math.sin(angle_deg * math.pi / 180)
    return np.array([[a_, b_, (1 - a_) * center[0] - b_ * center[1]], [-b_, a_, b_ * center[0] + (1 - a_) * center[1]]], np.float64)


def transform_points(points, transform) -> np.ndarray:
    p = np.c_[np.asarray(points, np.float64), np.ones(len(points))]
    return (np.asarray(transform, np.float64) @ p.T).T[:, :3]


def read_vertices(path: str) -> np.ndarray:
    """Vertices of a .ply (ascii or binary_little_endian, float/double x y z first among the vertex properties) or .obj file."""
    if path.lower().endswith(".obj"):
        v = [[float(t) for t in line.split()[1:4]] for line in open(path) if line.startswith("v ")]
        return np.asarray(v, np.float64)
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                props.append((tok[1], tok[-1]))
            elif tok[0] == "end_header":
                break
        names = [p[1] for p in props]
        ix = [names.index(a) for a in ("x", "y", "z")]
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(count)]
            return np.asarray([[float(r[i]) for i in ix] for r in rows], np.float64)
        if fmt != "binary_little_endian":
            raise ValueError("%s: unsupported PLY format %s" % (path, fmt))
        types = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
                 "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}
        dt = np.dtype([(n, types[t]) for t, n in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float64)


def _bbox_corners(v: np.ndarray) -> np.ndarray:
    lo, hi = v.min(0), v.max(0)
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float64)


def _shared_random_seed() -> int:
    """A fresh random seed that is the SAME on every replica: rank 0 draws it and broadcasts it when a process group exists."""
    seed = int(np.random.SeedSequence().entropy % (1 << 31))
    try:
        import torch.distributed as dist
    except ImportError:  # no torch.distributed in this interpreter: a single process keeps its own draw
        return seed
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # a failing collective must NOT be swallowed: ranks that kept their own seeds would draw different epoch permutations and their
        # shard slices would no longer partition the global batch (and the other ranks would hang in the broadcast)
        box = [seed]
        dist.broadcast_object_list(box, src=0)
        seed = int(box[0])
    return seed


def load_split(path: str, ratio: float, rng: np.random.Generator) -> List[int]:
    """_split_settings.json (utils/dataset_utils.py:462-497): reuse a stored split with the same ratio, else draw and store one."""
    fp = os.path.join(path, "_split_settings.json")
    if os.path.isfile(fp):
        info = json.load(open(fp))
        if info["split"][0]["ratio"] == ratio:
            return info["split"][0]["values"]
    n = len(glob.glob(path + "/*seg.png"))
    split = np.zeros(n, int)
    split[: int(n * ratio)] = 1
    rng.shuffle(split)
    json.dump({"split": [{"ratio": ratio, "values": split.tolist()}]}, open(fp, "w"))
    return split.tolist()


# ------------------------------------------------------------------------------------------------
class VectorfieldDataset:
    def __init__(self, root, path_meshes, no_points=9, color_input=False, normal=(0.5, 0.5), test=False, objectsofinterest=(), save=False, noise=2,
                 data_size=None, random_translation=(25.0, 25.0), random_rotation=15.0, random_crop=True, contrast=0.2, brightness=0.2, hue=0.05,
                 saturation=0.2, use_train_split=False, use_validation_split=False, train_validation_split=0.9, output_folder="", use_imgaug=False,
                 visibility_filter=False, separated_vectorfields=False, wxyz_quaterion_input=False, path_filter_root=None, seed: int = 0):
        if use_imgaug:
            raise NotImplementedError("use_imgaug pipelines (augmentation_model.py) are not part of this build")
        if separated_vectorfields:
            raise NotImplementedError("separated vector fields (modelname pvnet) are not part of this build")
        self.path_meshes, self.no_points, self.color_input, self.normal = path_meshes, no_points, color_input, list(normal)
        self.objectsofinterest = list(objectsofinterest)
        self.noise, self.random_translation, self.random_rotation, self.random_crop = noise, random_translation, random_rotation, random_crop
        self.contrast, self.brightness = contrast, brightness
        self.use_train_split, self.use_validation_split, self.train_validation_split = use_train_split, use_validation_split, train_validation_split
        self.visibility_filter, self.wxyz_quaterion_input = visibility_filter, wxyz_quaterion_input
        if seed is None:
            seed = _shared_random_seed()   # the reference shuffles / augments differently on every run; all replicas must still agree on ONE stream
        self.seed = int(seed)
        self.rng = np.random.default_rng(self.seed)   # re-seeded per image in generate_dataset(): the augmentation of image i of epoch e does
        #                                               not depend on the number of replicas or on which of them reads it
        self.order_rng = np.random.default_rng([self.seed, 1])  # epoch permutations only: identical on every replica
        self.meshes = self.load_meshes(path_meshes)
        self.imgs: List[Tuple[str, str, str, str, str]] = []
        self.class_labels: Dict[str, Dict[str, int]] = {}
        self.fixed_transformations: Dict[str, Dict[str, np.ndarray]] = {}
        self.camera_data: Dict[str, np.ndarray] = {}
        for name in sorted(os.listdir(str(root))):
            if path_filter_root is None or name in path_filter_root:
                self._explore(os.path.join(root, name))
        if not self.imgs:
            raise FileNotFoundError("no <name>.png + <name>.seg.png + <name>.json triples under %s" % root)

    def __len__(self):
        return len(self.imgs)

    # ---- discovery ---------------------------------------------------------------------------------
    def load_meshes(self, path) -> Dict[str, dict]:
        meshes = {}
        info_file = os.path.join(path, "models_info.json")
        info = json.load(open(info_file)) if os.path.isfile(info_file) else {}
        for name in sorted(o for o in os.listdir(path) if os.path.isdir(os.path.join(path, o))):
            model = os.path.join(path, name, name + ".obj")
            if not os.path.exists(model):
                model = os.path.join(path, name, name + ".ply")
            kp = os.path.join(path, name, name + "_keypoints.ply")
            if os.path.isfile(model) and os.path.isfile(kp):
                v = read_vertices(model)
                diam = info.get(name, {}).get("diameter")
                if diam is None:  # largest vertex distance (:632-640)
                    sub = v if len(v) <= 4000 else v[np.random.default_rng(0).choice(len(v), 4000, replace=False)]
                    G = sub @ sub.T
                    d2 = np.diag(G)[:, None] + np.diag(G)[None, :] - 2 * G
                    diam = float(np.sqrt(d2.max()))
                meshes[name] = {"keypoints": read_vertices(kp), "vertices": v, "volume": _bbox_corners(v), "diameter": float(diam)}
        return meshes

    def _explore(self, path):
        if not os.path.isdir(path):
            return
        sub = [os.path.join(path, o) for o in sorted(os.listdir(path)) if os.path.isdir(os.path.join(path, o))]
        if sub:
            for s in sub:
                self._explore(s)
            return
        files = sorted(glob.glob(path + "/*seg.png"))
        if not files:
            return
        if self.use_train_split or self.use_validation_split:
            split = np.array(load_split(path, self.train_validation_split, self.rng), bool)
            files = [f for f, s in zip(files, split if self.use_train_split else ~split) if s]
        if path not in self.class_labels:
            data = json.load(open(os.path.join(path, "_object_settings.json")))
            self.class_labels[path] = {o["class"]: o["segmentation_class_id"] for o in data["exported_objects"]}
            self.fixed_transformations[path] = {o["class"]: np.array(o["fixed_model_transform"], np.float32).T for o in data["exported_objects"]}
            cam = json.load(open(os.path.join(path, "_camera_settings.json")))["camera_settings"][0]["intrinsic_settings"]
            self.camera_data[path] = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], np.float64)
        for seg in files:
            for ext in ("png", "bmp", "jpg"):
                img = seg.replace("seg.png", ext)
                js = seg.replace("seg.png", "json")
                if os.path.exists(img) and os.path.exists(js):
                    self.imgs.append((img, os.path.basename(img), js, seg, path))
                    break

    def load_json_minimal(self, path) -> dict:
        data = json.load(open(path))
        out = {"keypoints2d": [], "objectClasses": {}, "poses_quaternions": [], "poses_loc": [], "px_count_all": []}
        idx = 0
        for info in data["objects"]:
            if self.visibility_filter and not info["visibility"] > 0.10:
                continue
            out["objectClasses"].setdefault(info["class"], []).append(idx)
            out["px_count_all"].append(int(info.get("px_count_all", 0)))
            out["keypoints2d"].append([(p[0], p[1]) for p in info["keypoints_2d"]])
            out["poses_quaternions"].append(np.array(info["quaternion_xyzw"], np.float32))
            out["poses_loc"].append(np.array(info["location"], np.float32))
            idx += 1
        return out

    # ---- one sample ----------------------------------------------------------------------------------
    def apply_preprocessing(self, item, imagesize, cropratio) -> dict:
        from PIL import Image

        img_path, name, js, seg_path, path_raw = item
        data = self.load_json_minimal(js)
        img = Image.open(img_path)
        img = img.convert("RGB") if self.color_input else img.convert("L")
        seg = Image.open(seg_path).convert("L")
        width, height = img.size
        crop_h = round(float(height) * cropratio)
        crop_w = crop_h * (float(imagesize[1]) / float(imagesize[0]))
        out_h, out_w = int(crop_h), int(crop_w)
        scale = imagesize[0] / out_h
        if self.random_crop:
            w_crop = int(self.rng.integers(0, width - out_w + 1))
            h_crop = int(self.rng.integers(0, height - out_h + 1))
        else:
            w_crop, h_crop = int((width - out_w) / 2), int((height - out_h) / 2)
        dx = round(self.rng.normal(0, 2) * float(self.random_translation[0]))
        dy = round(self.rng.normal(0, 2) * float(self.random_translation[1]))
        angle = round(self.rng.normal(0, 1) * float(self.random_rotation))
        offsets = np.array([h_crop, w_crop, out_h, out_w, dx, dy, angle, scale, width, height], np.float32)
        tm = np.array([[1, 0, dx], [0, 1, dy]], np.float64)
        rm = get_rotation_matrix_2D((width / 2, height / 2), angle)
        # output -> input map of the image warp (tfa.image.transform convention == PIL AFFINE)
        ar, at = np.identity(3), np.identity(3)
        ar[:2] = get_rotation_matrix_2D((width / 2, height / 2), -angle)
        at[:2] = [[1, 0, -dx], [0, 1, -dy]]
        affine = (ar @ at).flatten()[:6]
        if dx or dy or angle:
            img = img.transform(img.size, Image.AFFINE, data=tuple(affine), resample=Image.BILINEAR)
            seg = seg.transform(seg.size, Image.AFFINE, data=tuple(affine), resample=Image.NEAREST)
        box = (w_crop, h_crop, w_crop + out_w, h_crop + out_h)
        img = img.crop(box).resize((imagesize[1], imagesize[0]), Image.BILINEAR)
        seg = seg.crop(box).resize((imagesize[1], imagesize[0]), Image.NEAREST)
        img = np.asarray(img, np.float32)
        img = img[..., None] if img.ndim == 2 else img
        seg = np.asarray(seg, np.uint8)
        labels, fixed, cam = self.class_labels[path_raw], self.fixed_transformations[path_raw], self.camera_data[path_raw]
        oc, kp = len(self.objectsofinterest), self.no_points
        kp2 = np.full((oc, 1, kp, 2), -1000.0, np.float32)
        kp3 = np.zeros((oc, 1, kp, 3), np.float32)
        cub = np.zeros((oc, 1, 8, 3), np.float32)
        poses = np.zeros((oc, 1, 3, 4), np.float32)
        pxc = np.zeros((oc, 1, 1), np.float32)
        diam = np.full((oc, 1, 1), -1.0, np.float32)
        new_seg = np.zeros_like(seg)
        for o, obj in enumerate(self.objectsofinterest):
            mesh = self.meshes[obj]
            if obj in fixed:
                kp3[o, 0] = transform_points(mesh["keypoints"], fixed[obj])[:kp]
                cub[o, 0] = transform_points(mesh["volume"], fixed[obj])
            else:
                kp3[o, 0], cub[o, 0] = np.asarray(mesh["keypoints"])[:kp], mesh["volume"]
            for cls, ids in data["objectClasses"].items():
                if obj in cls:  # substring match like the reference (:379)
                    i = ids[0]
                    pts = np.asarray(data["keypoints2d"][i], np.float64)[:kp]
                    h1 = np.c_[pts, np.ones(len(pts))]
                    r3, t3 = np.identity(3), np.identity(3)
                    r3[:2], t3[:2] = rm, tm
                    p = (t3 @ (r3 @ h1.T))[:2].T - np.array([w_crop, h_crop])      # reproject (geometry_utils.py:7-19)
                    kp2[o, 0] = (p * scale)[:, ::-1]                                # stored (y,x) (:483)
                    poses[o, 0] = quaternion_matrix(data["poses_quaternions"][i], data["poses_loc"][i], wxyz_input=self.wxyz_quaterion_input)
                    pxc[o, 0, 0] = int(float(data["px_count_all"][i]) * scale + 0.5)
                    diam[o, 0, 0] = mesh["diameter"] * (np.linalg.norm(fixed[obj][:, 0]) if obj in fixed else 1.0)
                    new_seg[seg == labels[obj]] = o + 1
                    break
        # photometric part of image_augmentation (:259-272)
        if self.brightness:
            img = img + self.rng.uniform(-self.brightness, self.brightness)
        if self.contrast:
            f = self.rng.uniform(1 - self.contrast, 1 + self.contrast)
            m = img.mean(axis=(0, 1), keepdims=True)
            img = (img - m) * f + m
        img = ((img / 255.0) - self.normal[0]) / self.normal[1]
        if self.noise:
            img = img + self.rng.normal(0.0, self.rng.uniform(0, self.noise), img.shape)
        img = np.clip(img, -1, 1).astype(np.float32)
        if img.shape[2] == 1:
            img = np.repeat(img, 3, axis=2)
        p = os.path.normpath(path_raw.replace("\\", "/")).split(os.sep)
        return dict(img=img, label=new_seg, target_vert=kp2, keypoints3d=kp3, cam_mat=cam.astype(np.float32), diameters=diam, offsets=offsets,
                    cuboid3d=cub, poses_gt=poses, pixel_gt_count=pxc, image_id=p[-2] + "_" + p[-1] + "_" + os.path.splitext(name)[0])

    # ---- batches ---------------------------------------------------------------------------------------
    def generate_dataset(self, batchsize, epochs, prefetch=0, imagesize=(448, 448), cropratio=1.0, worker=1, no_objects=None, shuffle=True,
                         mirrored_strategy=None, shard: Tuple[int, int] = (0, 1)) -> Tuple[Iterator[Dict[str, torch.Tensor]], int]:
        """`batchsize` is the GLOBAL batch (vectorfield_dataset.py:923; `experimental_distribute_dataset` splits it, :1000-1002).  Here every
        replica owns a process: shard = (rank, world) makes it read, decode and augment only its contiguous slice of each global batch
        (the epoch permutation comes from a generator of its own, seeded alike on all replicas, so the slices partition the batch)."""
        from ..parallel import shard_range

        data_size = len(self.imgs) - (len(self.imgs) % batchsize)
        epoch_batches = data_size // batchsize
        oc = len(self.objectsofinterest)
        begin, end = shard_range(batchsize, shard[0], shard[1])

        def sample(epoch, i):
            self.rng = np.random.default_rng([self.seed, 2, epoch, int(i)])   # crop / rotation / colour draws of THIS image in THIS epoch
            return self.apply_preprocessing(self.imgs[i], imagesize, cropratio)

        def it():
            for epoch in range(max(int(epochs), 1)):
                order = self.order_rng.permutation(data_size) if shuffle else np.arange(data_size)
                for b in range(epoch_batches):
                    items = [sample(epoch, i) for i in order[b * batchsize + begin:b * batchsize + end]]
                    lab = np.stack([x["label"] for x in items])
                    st = lambda k: torch.from_numpy(np.stack([x[k] for x in items]))  # noqa: E731
                    yield dict(img=st("img"), target_seg=torch.from_numpy(np.eye(oc + 1, dtype=np.float32)[lab]), target_vert=st("target_vert"),
                               keypoints3d=st("keypoints3d"), cam_mat=st("cam_mat"), diameters=st("diameters"), offsets=st("offsets"),
                               filtered_seg=torch.from_numpy(lab[..., None].astype(np.int32)), cuboid3d=st("cuboid3d"), poses_gt=st("poses_gt"),
                               pixel_gt_count=st("pixel_gt_count"), image_id=[x["image_id"] for x in items])

        return it(), epoch_batches

    def generate_object_vertex_array(self):
        """(vertex_array [oc, Vmax, 3] in the fixed-transform frame, vertex_count [oc,1]) for ADD / ADD-S (:1047-1074)."""
        oc = len(self.objectsofinterest)
        count = np.zeros((oc, 1), np.int32)
        for i, o in enumerate(self.objectsofinterest):
            if o in self.meshes:
                count[i, 0] = len(self.meshes[o]["vertices"])
        arr = np.zeros((oc, int(count.max()), 3), np.float32)
        for i, o in enumerate(self.objectsofinterest):
            for fixed in self.fixed_transformations.values():
                if o in fixed and o in self.meshes:
                    arr[i, :count[i, 0]] = transform_points(self.meshes[o]["vertices"], fixed[o])
                    break
        return arr, count


# ------------------------------------------------------------------------------------------------
# exporter: the synthetic scene in the on-disk format above (round-trip tests; a sample dataset for the drivers)
# ------------------------------------------------------------------------------------------------
def write_ndds_scene(root: str, meshes_dir: str, scene, count: int, names: Sequence[str], scene_name: str = "000000"):
    """Render `count` images of a SyntheticSceneDataset (built for the full 480x640 frame) into <root>/<scene_name>/ and its
    object models into <meshes_dir>/ in the layout VectorfieldDataset reads."""
    from PIL import Image

    from .synthetic_scene import CAMERA, FULL_H, FULL_W

    assert scene.size == (FULL_H, FULL_W) and len(names) == scene.oc
    d = os.path.join(root, scene_name)
    os.makedirs(d, exist_ok=True)
    os.makedirs(meshes_dir, exist_ok=True)

    def write_ply(path, v):
        with open(path, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(v))
            for p in v:
                f.write("%.6f %.6f %.6f\n" % tuple(p))

    info = {}
    for o, name in enumerate(names):
        os.makedirs(os.path.join(meshes_dir, name), exist_ok=True)
        write_ply(os.path.join(meshes_dir, name, name + ".ply"), scene.mesh_vertex_array[o])
        write_ply(os.path.join(meshes_dir, name, name + "_keypoints.ply"), scene.keypoints3d[o])
        info[name] = {"diameter": float(scene.diameters[o])}
    json.dump(info, open(os.path.join(meshes_dir, "models_info.json"), "w"))
    json.dump({"exported_objects": [{"class": n, "segmentation_class_id": 10 * (o + 1), "fixed_model_transform": np.identity(4).tolist()}
                                    for o, n in enumerate(names)]}, open(os.path.join(d, "_object_settings.json"), "w"))
    json.dump({"camera_settings": [{"intrinsic_settings": {"fx": CAMERA[0, 0], "fy": CAMERA[1, 1], "cx": CAMERA[0, 2], "cy": CAMERA[1, 2]}}]},
              open(os.path.join(d, "_camera_settings.json"), "w"))
    for i in range(count):
        it = scene._render(np.random.default_rng([scene.seed, i]))
        img = np.clip((it["img"] * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(d, "%06d.png" % i))
        Image.fromarray((it["label"].astype(np.uint8) * 10)).save(os.path.join(d, "%06d.seg.png" % i))
        objs = []
        for o, name in enumerate(names):
            P = it["poses"][o]
            objs.append({"class": name, "visibility": 1.0, "px_count_all": int(it["counts"][o]), "location": P[:, 3].tolist(),
                         "quaternion_xyzw": matrix_to_quaternion_xyzw(P[:, :3]).tolist(),
                         "keypoints_2d": it["kp2"][o][:, ::-1].tolist(), "keypoints_3d": scene.keypoints3d[o].tolist()})
        json.dump({"objects": objs}, open(os.path.join(d, "%06d.json" % i), "w"))
