"""Model constructors with the reference's signatures (casapose/pose_models/models/pose_models.py)."""
from __future__ import annotations

from .model import CasaposeModel


def _conditional(name, partial, guided, ver_dim, seg_dim, *args, bilinear=(False,) * 5, sharing=None, **kwargs):
    return _conditional_impl(name, partial, guided, bilinear, sharing or {}, ver_dim, seg_dim, *args, **kwargs)


def _conditional_impl(name, partial, guided, bilinear, sharing, ver_dim, seg_dim, fcdim=256, s8dim=128, s4dim=64, s2dim=32, raw_dim=32, input_shape=None,
                 input_segmentation_shape=None, input_tensor=None, weights=None, base_model="resnet18",
                 backbone=None, output_lablemap=False, **kwargs):
    if base_model != "resnet18":
        raise TypeError("Undefined base model type: {}".format(base_model)) if base_model not in (
            "resnet34", "resnet50", "resnet101", "resnet152") else NotImplementedError(
            "backbone %s is not built for MI355X yet (resnet18 is)" % base_model)
    if backbone is not None or input_tensor is not None:
        raise NotImplementedError("external backbone / input_tensor are Keras-graph features without an equivalent here")
    return CasaposeModel(name, ver_dim, seg_dim, (fcdim, s8dim, s4dim, s2dim, raw_dim), input_shape=input_shape,
                         input_segmentation_shape=input_segmentation_shape, weights=weights,
                         output_lablemap=output_lablemap, device=kwargs.get("device"), seed=kwargs.get("seed"),
                         fuse_upsample=kwargs.get("fuse_upsample", True), fuse_heads=kwargs.get("fuse_heads", True), conv_mode=kwargs.get("conv_mode"), f16x2_guard=kwargs.get("f16x2_guard"),
                         partial=partial, guided=guided, bilinear=bilinear, **sharing)


_GU = (False, True, True, True, False)  # blocks 7, 8, 9 upsample with the label-guided gather


def CASAPoseConditional1(*args, **kwargs):
    """casapose_c (pose_models.py:14-129): CLADE in every decoder-2 block, ordinary convolutions, nearest x2 upsampling.
    The reference leaves the HalfSize 1x1 convolutions of this variant trainable (:59-61); they stay at their identity
    initialisation here (the conditioning is a hard label map)."""
    return _conditional("casapose_c", (False,) * 5, (False,) * 5, *args, **kwargs)


def CASAPoseConditional2(*args, **kwargs):
    """casapose_c_gu (pose_models.py:132-256): no partial convolution, guided upsampling."""
    return _conditional("casapose_c_gu", (False,) * 5, _GU, *args, **kwargs)


def CASAPoseConditional3(*args, **kwargs):
    """casapose_c_gcu3 (pose_models.py:259-383): partial convolution in blocks 6-8, guided upsampling."""
    return _conditional("casapose_c_gcu3", (True, True, True, False, False), _GU, *args, **kwargs)


def CASAPoseConditional4(*args, **kwargs):
    """casapose_c_gcu4 (pose_models.py:386-510): partial convolution in blocks 6-9, guided upsampling."""
    return _conditional("casapose_c_gcu4", (True, True, True, True, False), _GU, *args, **kwargs)


def CASAPoseConditional9(*args, **kwargs):
    """casapose_c_gcu4_bilat (pose_models.py:1102-1229): gcu4 with GuidedBilinearUpsampling instead of the guided nearest gather."""
    return _conditional("casapose_c_gcu4_bilat", (True, True, True, True, False), _GU, *args, bilinear=_GU, **kwargs)


def CASAPoseConditional6(*args, **kwargs):
    """casapose_c_gcu5_sw5 (pose_models.py:699-839): both decoders use the SAME five PartialConvolution weight sets
    (`pv_block_{i}_{i+5}_conv2d`): plain SAME convolutions in decoder 1, mask-aware in decoder 2; block 6 has no convolution -- it
    applies CLADE to the raw output of block 1's convolution (:731,:762-770)."""
    return _conditional("casapose_c_gcu5_sw5", (False, True, True, True, True), _GU, *args,
                        sharing=dict(shared=(True,) * 5, reuse_first=True), **kwargs)


def CASAPoseConditional7(*args, **kwargs):
    """casapose_c_gcu4_sw1 (pose_models.py:842-969): the decoders share the first convolution (and its output); blocks 7-10 are partial."""
    return _conditional("casapose_c_gcu4_sw1", (False, True, True, True, True), _GU, *args,
                        sharing=dict(shared=(True, False, False, False, False), reuse_first=True), **kwargs)


def CASAPoseConditional8(*args, **kwargs):
    """casapose_c_gcu5_sw1 (pose_models.py:972-1099): like _sw1 above, but decoder 2 takes NO skip connections (:1033-1079 feed y alone)."""
    return _conditional("casapose_c_gcu5_sw1", (False, True, True, True, True), _GU, *args,
                        sharing=dict(shared=(True, False, False, False, False), reuse_first=True, skips2=False), **kwargs)


def CASAPoseConditional10(*args, **kwargs):
    """casapose_c_gcu4_sw2 (pose_models.py:1232-1362): blocks 1/6 and 2/7 share their weights (decoder 2 applies them mask-aware to its
    own inputs); blocks 8, 9 partial, block 10 an ordinary convolution."""
    return _conditional("casapose_c_gcu4_sw2", (True, True, True, True, False), _GU, *args,
                        sharing=dict(shared=(True, True, False, False, False)), **kwargs)


def CASAPoseConditional5(*args, **kwargs):
    """casapose_c_gcu5: ResNet-18 (OS 8) + segmentation decoder + class-adaptive vector-field
    decoder with 5 partial convolutions, CLADE and guided upsampling (pose_models.py:513-635)."""
    return _conditional("casapose_c_gcu5", (True,) * 5, _GU, *args, **kwargs)


def PVNet(ver_dim, seg_dim, fcdim=256, s8dim=128, s4dim=64, s2dim=32, raw_dim=32, input_shape=None, input_tensor=None, weights=None,
          base_model="resnet18", backbone=None, output_lablemap=False, **kwargs):
    """pvnet_combined (pose_models.py:645-696): the baseline without the class-adaptive decoder -- ResNet-18 + decoder 1 + one 1x1 head
    `pv_final_conv` with seg_dim + ver_dim output channels.  The registry key `pvnet` is the same graph with per-object ("separated")
    vector fields, ver_dim = 2*points*objects (train_casapose.py:221,313-320): inference and training are built for both forms -- the merged
    output through cp_pose_loss_f32, the separated fields through cp_pose_loss_sep_f32 (compute_loss's per-object branch,
    train_casapose.py:57,97-125; no keypoint loss there, as in the reference, whose voter reads merged fields)."""
    if base_model != "resnet18":
        raise NotImplementedError("backbone %s is not built for MI355X yet (resnet18 is)" % base_model)
    if backbone is not None or input_tensor is not None:
        raise NotImplementedError("external backbone / input_tensor are Keras-graph features without an equivalent here")
    return CasaposeModel("pvnet_combined", ver_dim, seg_dim, (fcdim, s8dim, s4dim, s2dim, raw_dim), input_shape=input_shape, weights=weights,
                         output_lablemap=output_lablemap, device=kwargs.get("device"), seed=kwargs.get("seed"),
                         fuse_upsample=kwargs.get("fuse_upsample", True), fuse_heads=False, pvnet=True)
