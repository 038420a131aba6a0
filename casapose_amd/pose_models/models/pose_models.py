"""Model constructors with the reference's signatures (casapose/pose_models/models/pose_models.py)."""
from __future__ import annotations

from .model import CasaposeModel


def CASAPoseConditional5(ver_dim, seg_dim, fcdim=256, s8dim=128, s4dim=64, s2dim=32, raw_dim=32, input_shape=None,
                         input_segmentation_shape=None, input_tensor=None, weights=None, base_model="resnet18",
                         backbone=None, output_lablemap=False, **kwargs):
    """casapose_c_gcu5: ResNet-18 (OS 8) + segmentation decoder + class-adaptive vector-field
    decoder with 5 partial convolutions, CLADE and guided upsampling (pose_models.py:513-635)."""
    if base_model != "resnet18":
        raise TypeError("Undefined base model type: {}".format(base_model)) if base_model not in (
            "resnet34", "resnet50", "resnet101", "resnet152") else NotImplementedError(
            "backbone %s is not built for MI355X yet (resnet18 is)" % base_model)
    if backbone is not None or input_tensor is not None:
        raise NotImplementedError("external backbone / input_tensor are Keras-graph features without an equivalent here")
    return CasaposeModel("casapose_c_gcu5", ver_dim, seg_dim, (fcdim, s8dim, s4dim, s2dim, raw_dim), input_shape=input_shape,
                         input_segmentation_shape=input_segmentation_shape, weights=weights,
                         output_lablemap=output_lablemap, device=kwargs.get("device"), seed=kwargs.get("seed"),
                         fuse_upsample=kwargs.get("fuse_upsample", True), fuse_heads=kwargs.get("fuse_heads", True))
