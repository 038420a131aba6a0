"""Command-line / INI configuration with the reference's flag set and precedence
(casapose/utils/config_parser.py:7-170): argparse default < INI `[defaults]` < command line.

Table-driven re-implementation on the standard library only (configparser.SafeConfigParser,
used by the reference, no longer exists).  `parse_config(argv=None)` returns the same
`argparse.Namespace` the reference's scripts consume, including the post-processing of
image sizes, comma lists, `objects_to_copy`, output folders and the random seed.
"""
from __future__ import annotations

import argparse
import configparser
from typing import List, Optional, Sequence

import numpy as np

_TRUE = ("yes", "true", "t", "y", "1")
_FALSE = ("no", "false", "f", "n", "0")


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).lower()
    if s in _TRUE:
        return True
    if s in _FALSE:
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


# (flag, type, default, nargs) -- type None = plain string
_B, _I, _F, _S = str2bool, int, float, None
FLAGS = [
    # data
    ("data", _S, "", None), ("data_path_filter", _S, None, None), ("datatest", _S, "", None),
    ("datatest_path_filter", _S, None, None), ("color_dataset", _B, True, None),
    ("data_wxyz_quaterion", _B, False, None), ("datatest_wxyz_quaterion", _B, False, None),
    ("datameshes", _S, "", None), ("train_validation_split", _F, 0.9, None),
    # model
    ("modelname", _S, "casapose_cond_weighted", None), ("backbonename", _S, "resnet18", None),
    ("estimate_confidence", _B, False, None), ("estimate_coords", _B, False, None),
    ("confidence_regularization", _B, False, None), ("confidence_filter_estimates", _B, True, None),
    ("confidence_choose_second", _B, False, None), ("object", _S, None, None), ("no_points", _I, 9, None),
    ("pretrained", _B, True, None), ("train_vectors_with_ground_truth", _B, False, None),
    # losses
    ("mask_loss_weight", _F, 1.0, None), ("vertex_loss_weight", _F, 0.5, None), ("proxy_loss_weight", _F, 0.013, None),
    ("keypoint_loss_weight", _F, 0.0, None), ("filter_vertex_with_segmentation", _B, False, None),
    ("filter_high_proxy_errors", _B, False, None), ("use_bpnp_reprojection_loss", _B, False, None),
    ("max_keypoint_pixel_error", _F, 25.0, None),
    # optimisation
    ("workers", _I, 1, None), ("prefetch", _I, 0, None), ("batchsize", _I, 32, None),
    ("imagesize", _I, [448], "+"), ("imagesize_test", _I, [448], "+"), ("crop_factor", _F, 1.0, None),
    ("lr", _F, 0.001, None), ("lr_decay", _F, 1.0, None), ("lr_epochs", _I, 15, None), ("lr_epochs_start", _I, 0, None),
    ("lr_epochs_steps", _S, None, None), ("epochs", _I, 60, None), ("gpuids", _I, [0], "+"), ("manualseed", _I, None, None),
    # augmentation
    ("noise", _F, 0.0, None), ("contrast", _F, 0.4, None), ("brightness", _F, 0.2, None), ("saturation", _F, 0.001, None),
    ("hue", _F, 0.001, None), ("use_imgaug", _B, False, None), ("rotation", _F, 15, None), ("translation", _F, 25, None),
    # logging / evaluation
    ("loginterval", _I, 100, None), ("saveinterval", _I, 10, None), ("validationinterval", _I, 1, None),
    ("save_debug_batch", _B, False, None), ("save_eval_batches", _B, False, None), ("write_poses", _B, False, None),
    ("filter_test_with_gt", _B, False, None), ("min_object_size_test", _I, 1, None),
    ("net", _S, "./output/training_checkpoints", None), ("outf", _S, "tmp", None), ("evalf", _S, "", None),
    # weights
    ("load_h5_weights", _B, False, None), ("load_h5_filename", _S, "result_w", None),
    ("copy_weights_from_backup_network", _B, False, None), ("copy_weights_add_confidence_maps", _B, False, None),
    ("objects_to_copy", _I, 0, None), ("objects_in_input_network", _I, 0, None), ("objects_to_copy_list", _S, "", None),
]
_INT_LISTS = ("gpuids", "imagesize", "imagesize_test")


def _csv(val: Optional[str]) -> Optional[List[str]]:
    return None if val is None else [x.strip() for x in val.split(",")]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    for name, typ, default, nargs in FLAGS:
        kw = {"default": default}
        if typ is not None:
            kw["type"] = typ
        if nargs is not None:
            kw["nargs"] = nargs
        parser.add_argument("--" + name, **kw)
    return parser


def parse_config(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("-c", "--config", metavar="FILE")
    known, rest = pre.parse_known_args(argv)
    parser = build_parser()
    if known.config:
        ini = configparser.ConfigParser(allow_no_value=True)
        if not ini.read([known.config]):
            raise FileNotFoundError(known.config)
        overrides = dict(ini.items("defaults"))
        for key in _INT_LISTS:
            if key in overrides:
                overrides[key] = [int(t) for t in overrides[key].split(",")]
        # string values from the INI still go through each flag's `type` (argparse applies the
        # type to string defaults), exactly like the reference's parser.set_defaults(**defaults)
        parser.set_defaults(**overrides)
    opt = parser.parse_args(rest)

    for key in ("imagesize", "imagesize_test"):
        v = getattr(opt, key)
        setattr(opt, key, (v[0], v[0]) if len(v) == 1 else (v[0], v[1]))
    opt.data_path_filter = _csv(opt.data_path_filter)
    opt.datatest_path_filter = _csv(opt.datatest_path_filter)
    if opt.lr_epochs_steps is not None:
        opt.lr_epochs_steps = [int(x) for x in _csv(opt.lr_epochs_steps)]
    if opt.objects_to_copy_list == "":
        idx = np.arange(opt.objects_to_copy + 1, dtype=np.int32)
        opt.objects_to_copy = np.stack([idx, idx], axis=1)
    else:
        table = np.array(np.genfromtxt(opt.objects_to_copy_list, delimiter=","), np.int32).reshape(-1, 2)
        opt.objects_to_copy = np.concatenate((np.array([[0, 0]], np.int32), table))  # row 0 = background
    if opt.objects_in_input_network == 0:
        opt.objects_in_input_network = opt.objects_to_copy.shape[0] - 1
    if opt.evalf == "":
        opt.evalf = opt.outf
    if "/" not in opt.outf:
        opt.outf = "output/{}".format(opt.outf)
    if "/" not in opt.evalf:
        opt.evalf = opt.outf + "/" + opt.evalf
    if opt.manualseed is None:
        opt.manualseed = int(np.random.randint(1, 10000))
    return opt
