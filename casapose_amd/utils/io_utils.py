"""Pose files of an evaluation run, in the reference's on-disk formats (casapose/utils/io_utils.py:54-138, called from
test_casapose.py:392-399 when `--write_poses` is set):

  <path_out>/bop_evaluation.csv                     BOP-challenge result rows `scene_id,im_id,obj_id,score,R,t,time`, one per
                                                    object that is present in the ground truth of the image
  <path_out>/all_poses/poses_init_<name>.txt        every estimate (also false positives), 12 numbers per line
  <path_out>/filtered_poses/poses_gt_<name>.txt     ground truth, zeros where the object is absent
  <path_out>/filtered_poses/poses_init_<name>.txt   estimates, zeros where the object is absent in the ground truth

Host code: strings and a handful of floats per image.
"""
from __future__ import annotations

import os
import re
from typing import Optional, Sequence

import numpy as np

_NUMBER = re.compile(r"\d*\.*\d+")  # the reference's pattern (io_utils.py:60,125)
BOP_HEADER = "scene_id,im_id,obj_id,score,R,t,time\n"
POSE_HEADER = "#r11 r12 r13 r21 r22 r23 r31 r32 r33 tx ty tz\n"


def _text(x) -> str:
    while isinstance(x, (list, tuple, np.ndarray)):
        x = x[0]
    if hasattr(x, "numpy"):
        x = x.numpy()
        return _text(x)
    return x.decode("utf-8") if isinstance(x, (bytes, np.bytes_)) else str(x)


def _pose(p) -> np.ndarray:
    p = p.detach().cpu().numpy() if hasattr(p, "detach") else np.asarray(p)
    return p.astype(np.float32).reshape(3, 4)


def _numbers(values) -> str:
    return " ".join(map(str, values))


def _append(path: str, header: str, line: str):
    exists = os.path.isfile(path)
    with open(path, "a") as f:
        if not exists:
            f.write(header)
        f.write(line)


def write_poses(gt_poses, estimated_poses, names: Sequence[str], image_id, path_out: str, time_needed: Optional[float] = None):
    """gt_poses [objects,1,3,4] (or [objects,3,4]), estimated_poses [objects,3,4], names = `objectsofinterest`, image_id =
    the batch tuple's entry 12 (`<dir>_<dir>_<file stem>`: its first number is the scene, its second the image).
    score = 1 for a non-zero estimate, 0 for "not found" (all-zero pose); time = -1 when not measured."""
    gt = np.asarray(gt_poses.detach().cpu() if hasattr(gt_poses, "detach") else gt_poses, np.float32)
    if gt.ndim == 4:
        gt = gt[:, 0]
    found = _NUMBER.findall(_text(image_id))
    scene_id, img_id = int(found[0]), int(found[1])
    time = -1.0 if time_needed is None else float(time_needed)
    all_dir, filtered_dir = path_out + "all_poses/", path_out + "filtered_poses/"
    for d in (path_out, all_dir, filtered_dir):
        os.makedirs(d, exist_ok=True)

    def pose_line(pose):
        return _numbers(pose[:, :3].reshape(-1)) + " " + _numbers(pose[:, 3].reshape(-1)) + "\n"

    zeros = np.zeros((3, 4), np.float32)
    for idx, name in enumerate(names):
        obj_id = int(_NUMBER.findall(name)[0])
        est = _pose(estimated_poses[idx])
        if abs(float(gt[idx].sum())) > 0.0001:
            score = 1.0 if abs(float(est.sum())) > 0 else 0.0
            row = "%d,%d,%d,%s,%s,%s,%s\n" % (scene_id, img_id, obj_id, str(score), _numbers(est[:, :3].reshape(-1)),
                                               _numbers(est[:, 3].reshape(-1)), str(time))
            _append(path_out + "bop_evaluation.csv", BOP_HEADER, row)
            _append(filtered_dir + "poses_gt_" + name + ".txt", POSE_HEADER, pose_line(_pose(gt[idx])))
            _append(filtered_dir + "poses_init_" + name + ".txt", POSE_HEADER, pose_line(est))
        else:
            _append(filtered_dir + "poses_gt_" + name + ".txt", POSE_HEADER, pose_line(zeros))
            _append(filtered_dir + "poses_init_" + name + ".txt", POSE_HEADER, pose_line(zeros))
        _append(all_dir + "poses_init_" + name + ".txt", POSE_HEADER, pose_line(est))


def latest_checkpoint(checkpoint_path: str):
    """(path, number) of the newest `ckpt-<n>.npz` of a run, or None -- tf.train.latest_checkpoint for the checkpoints
    train_casapose.py writes (train_casapose.py:350,393,901)."""
    import glob

    found = []
    for p in glob.glob(os.path.join(checkpoint_path, "ckpt-*.npz")):
        try:
            found.append((int(os.path.basename(p)[5:-4]), p))
        except ValueError:
            pass
    if not found:
        return None
    n, p = max(found)
    return p, n
