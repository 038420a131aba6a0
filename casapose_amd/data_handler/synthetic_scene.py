"""A self-contained stand-in for the NDDS/BOP image source of the reference (VectorfieldDataset,
casapose/data_handler/vectorfield_dataset.py): ray-cast scenes of textured ellipsoids with analytic ground truth,
delivered as the SAME batch tuple the training / evaluation steps consume (train_casapose.py:496-507):

    img [B,H,W,3] in [-1,1], target_seg [B,H,W,K] one-hot, keypoints3d [B,oc,1,kp,3], target_vert [B,oc,1,kp,2] 2-D
    keypoints (y,x) in crop pixels, cam_mat [B,3,3], diameters [B,oc,1], offsets [B,10], filtered_seg [B,H,W,1] label
    map, poses_gt [B,oc,1,3,4], pixel_gt_count [B,oc]

There is no dataset on this machine (no network); this generator is what `--data synthetic[:N]` selects in
train_casapose.py / test_casapose.py so that the whole config-driven pipeline (losses, voting, PnP, ADD metrics, CSV
logs) runs end to end with ground truth that is exact by construction.  The NDDS reader itself is the next row of
SURVEY 8(f).

Geometry: object o is the ellipsoid x^2/a^2 + y^2/b^2 + z^2/c^2 = 1 in its own frame; its 9 keypoints are the centre and
the 8 corners of its bounding box (the reference's farthest-point keypoints are likewise spread over the object); the
evaluation mesh is a Fibonacci sampling of the surface.  A camera with the LINEMOD intrinsics renders 480x640; crops
follow the reference's offsets convention [h_crop, w_crop, -, -, dx, dy, angle, scale, 640, 480].
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch

CAMERA = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]])
FULL_H, FULL_W = 480, 640


def _rot(rng) -> np.ndarray:
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _fibonacci_sphere(n: int) -> np.ndarray:
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    th = np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], axis=1)


class SyntheticSceneDataset:
    def __init__(self, no_objects: int, image_size: Tuple[int, int] = (448, 448), no_points: int = 9, length: int = 64, seed: int = 0,
                 mesh_vertices: int = 642, random_crop: bool = True):
        if no_points != 9:
            raise ValueError("the synthetic scene defines 9 keypoints per object (centre + bounding-box corners)")
        self.oc, self.size, self.kp, self.length, self.seed = no_objects, tuple(image_size), no_points, length, seed
        self.random_crop = random_crop
        rng = np.random.default_rng(seed)
        self.axes = rng.uniform(35.0, 70.0, (no_objects, 3))                      # semi-axes, mm
        self.colors = rng.uniform(0.15, 0.95, (no_objects, 3))
        corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64)
        self.keypoints3d = np.concatenate([np.zeros((no_objects, 1, 3)), corners[None] * self.axes[:, None, :]], axis=1)  # [oc,9,3]
        sph = _fibonacci_sphere(mesh_vertices)
        self.mesh_vertex_array = sph[None] * self.axes[:, None, :]                  # [oc,V,3]
        self.mesh_vertex_count = np.full((no_objects, 1), mesh_vertices, np.int32)
        self.diameters = 2.0 * self.axes.max(axis=1)                                 # largest vertex distance

    def __len__(self):
        return self.length

    # ---- one image ------------------------------------------------------------------------------------
    def _render(self, rng) -> Dict[str, np.ndarray]:
        oc = self.oc
        H, W = self.size
        if (H, W) == (FULL_H, FULL_W):
            hc, wc = 0, 0
        elif self.random_crop:
            hc, wc = int(rng.integers(0, FULL_H - H + 1)), int(rng.integers(0, FULL_W - W + 1))
        else:
            hc, wc = (FULL_H - H) // 2, (FULL_W - W) // 2
        K = CAMERA
        # poses: centres spread over the crop, in front of the camera
        poses = np.zeros((oc, 3, 4))
        for o in range(oc):
            z = rng.uniform(650.0, 1100.0)
            u = rng.uniform(wc + 0.12 * W, wc + 0.88 * W)
            v = rng.uniform(hc + 0.12 * H, hc + 0.88 * H)
            poses[o, :, :3] = _rot(rng)
            poses[o, :, 3] = [(u - K[0, 2]) * z / K[0, 0], (v - K[1, 2]) * z / K[1, 1], z]
        # ray casting on the crop
        ys, xs = np.mgrid[hc:hc + H, wc:wc + W].astype(np.float64) + 0.5
        rays = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], axis=-1)  # camera frame
        depth = np.full((H, W), np.inf)
        label = np.zeros((H, W), np.uint8)
        shade = np.zeros((H, W))
        for o in range(oc):
            R, t = poses[o, :, :3], poses[o, :, 3]
            inv = 1.0 / self.axes[o]
            d = (rays @ R) * inv           # ray direction in the unit-sphere frame (R^T r, scaled)
            c = (-(R.T @ t)) * inv         # camera centre in that frame
            a = (d * d).sum(-1)
            bq = (d * c).sum(-1)
            cq = c @ c - 1.0
            disc = bq * bq - a * cq
            hit = disc > 0
            s = np.where(hit, (-bq - np.sqrt(np.where(hit, disc, 0.0))) / a, np.inf)  # ray parameter = depth (r_z = 1)
            nearer = hit & (s > 0) & (s < depth)
            nrm = (c + d * s[..., None])                                                 # unit-sphere normal
            depth = np.where(nearer, s, depth)
            label = np.where(nearer, o + 1, label).astype(np.uint8)
            shade = np.where(nearer, 0.55 + 0.45 * np.abs(nrm[..., 2]), shade)
        img = np.empty((H, W, 3))
        bg = 0.35 + 0.1 * np.sin(xs / 37.0)[..., None] * np.cos(ys / 53.0)[..., None] + rng.normal(0, 0.02, (H, W, 3))
        col = np.concatenate([np.zeros((1, 3)), self.colors])[label]
        img[:] = np.where(label[..., None] > 0, col * shade[..., None], bg)
        img = np.clip(img + rng.normal(0, 0.01, img.shape), 0, 1) * 2.0 - 1.0        # normal = [0.5, 0.5] -> [-1, 1]
        # 2-D keypoints in crop pixels, (y,x)
        kp2 = np.zeros((oc, self.kp, 2))
        for o in range(oc):
            cam = self.keypoints3d[o] @ poses[o, :, :3].T + poses[o, :, 3]
            uv = (cam @ K.T)
            uv = uv[:, :2] / uv[:, 2:]
            kp2[o, :, 0], kp2[o, :, 1] = uv[:, 1] - hc, uv[:, 0] - wc
        counts = np.array([(label == o + 1).sum() for o in range(oc)], np.int32)
        return dict(img=img.astype(np.float32), label=label, poses=poses, kp2=kp2, counts=counts,
                    offsets=np.array([hc, wc, 0, 0, 0, 0, 0, 1, FULL_W, FULL_H], np.float32))

    def batch(self, index: int, batchsize: int, shard: Tuple[int, int] = (0, 1)) -> Dict[str, torch.Tensor]:
        """Deterministic batch `index` (images index*batchsize ... of the endless stream seeded by `seed`).  `shard = (rank, world)`
        renders only that replica's contiguous slice of the global batch (casapose_amd.parallel.shard_range) -- image i of the
        stream is the same picture whichever replica draws it."""
        from ..parallel import shard_range

        begin, end = shard_range(batchsize, shard[0], shard[1])
        items = [self._render(np.random.default_rng([self.seed, index * batchsize + i])) for i in range(begin, end)]
        K1 = self.oc + 1
        lab = np.stack([it["label"] for it in items])
        seg = np.eye(K1, dtype=np.float32)[lab]
        b = end - begin
        return dict(
            img=torch.from_numpy(np.stack([it["img"] for it in items])),
            target_seg=torch.from_numpy(seg),
            keypoints3d=torch.from_numpy(np.tile(self.keypoints3d[None, :, None], (b, 1, 1, 1, 1)).astype(np.float32)),
            target_vert=torch.from_numpy(np.stack([it["kp2"] for it in items])[:, :, None].astype(np.float32)),
            cam_mat=torch.from_numpy(np.tile(CAMERA[None], (b, 1, 1)).astype(np.float32)),
            diameters=torch.from_numpy(np.tile(self.diameters[None, :, None, None], (b, 1, 1, 1)).astype(np.float32)),
            offsets=torch.from_numpy(np.stack([it["offsets"] for it in items])),
            filtered_seg=torch.from_numpy(lab[..., None].astype(np.int32)),
            poses_gt=torch.from_numpy(np.stack([it["poses"] for it in items])[:, :, None].astype(np.float32)),
            pixel_gt_count=torch.from_numpy(np.stack([it["counts"] for it in items]).astype(np.float32)[:, :, None, None]),
        )

    def generate_object_vertex_array(self):
        """(vertex_array [oc,V,3], vertex_count [oc,1]) like VectorfieldDataset.generate_object_vertex_array."""
        return self.mesh_vertex_array.astype(np.float32), self.mesh_vertex_count

    def generate_dataset(self, batchsize: int, epochs: int = 1, *_unused, shard: Tuple[int, int] = (0, 1), **_unused_kw) -> Tuple[Iterator[Dict[str, torch.Tensor]], int]:
        """(iterator over epochs*batches batches, batches per epoch) -- the shape of VectorfieldDataset.generate_dataset
        (vectorfield_dataset.py:905-1013).  The same `length` images are revisited every epoch.  `batchsize` is the GLOBAL batch
        (vectorfield_dataset.py:923,1002); with shard = (rank, world) every replica produces only its own slice of each batch."""
        n = self.length // batchsize

        def it():
            for _ in range(max(epochs, 1) + 1):
                for i in range(n):
                    yield self.batch(i, batchsize, shard)

        return it(), n
