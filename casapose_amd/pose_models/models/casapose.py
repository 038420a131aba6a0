"""Generic builder surface (casapose/pose_models/models/casapose.py:18-39,145-163)."""
from __future__ import annotations

import collections

from .model import CasaposeModel

DecoderParams = collections.namedtuple(
    "DecoderParams", ["weighted_clade", "partial_conv", "guided_upsampling", "bilinear_upsampling", "reuse_conv"]
)

# decoder-2 configuration of the five blocks "6".."10" (casapose.py:27-35)
CASAPOSE_PARAMS = {
    "clade": [
        DecoderParams(True, True, False, False, False),
        DecoderParams(True, True, True, False, False),
        DecoderParams(True, True, True, False, False),
        DecoderParams(True, True, True, False, False),
        DecoderParams(True, True, False, False, False),
    ],
}

_GCU5 = [tuple(p) for p in CASAPOSE_PARAMS["clade"]]


def CASAPose(layer_params, ver_dim, seg_dim, fcdim=256, s8dim=128, s4dim=64, s2dim=32, raw_dim=32, input_shape=None,
             input_segmentation_shape=None, input_tensor=None, weights=None, base_model="resnet18", backbone=None,
             output_lablemap=False, learn_upsampling=False, **kwargs):
    params = [DecoderParams(*p) for p in layer_params]
    if len(params) != 5:
        raise ValueError("layer_params must describe the five decoder blocks")
    if learn_upsampling or any((p.bilinear_upsampling and not p.guided_upsampling) or not p.weighted_clade for p in params):
        raise NotImplementedError("built for MI355X: weighted_clade=True with any combination of partial_conv / guided_upsampling / reuse_conv (+ "
                                  "bilinear_upsampling = GuidedBilinearUpsampling on guided blocks); unguided bilinear upsampling in decoder 2, plain "
                                  "ClassAdaptiveNormalization (its gather_nd on the float mask does not execute in the reference either) and "
                                  "learn_upsampling are not")
    if base_model != "resnet18":
        raise NotImplementedError("backbone %s is not built for MI355X yet" % base_model)
    return CasaposeModel("casapose_custom", ver_dim, seg_dim, (fcdim, s8dim, s4dim, s2dim, raw_dim), input_shape=input_shape,
                         input_segmentation_shape=input_segmentation_shape, weights=weights, output_lablemap=output_lablemap,
                         device=kwargs.get("device"), seed=kwargs.get("seed"), fuse_upsample=kwargs.get("fuse_upsample", True), fuse_heads=kwargs.get("fuse_heads", True), conv_mode=kwargs.get("conv_mode"), f16x2_guard=kwargs.get("f16x2_guard"),
                         # reuse_conv (casapose.py:178-197,236-237,260): block i+1 and block i+6 share ONE one-input PartialConvolution
                         # `pv_block_{i+1}_{i+6}_conv2d` -- an ordinary SAME convolution on both sides (`partial_conv and not reuse_conv`, :261);
                         # with reuse_conv on block 1, block 6 normalises block 1's raw convolution output instead of convolving (:188-190,236)
                         partial=[p.partial_conv and not p.reuse_conv for p in params], guided=[p.guided_upsampling for p in params],
                         bilinear=[p.bilinear_upsampling for p in params], shared=[p.reuse_conv for p in params], reuse_first=bool(params[0].reuse_conv))


def CASAPoseConditional(*args, **kwargs):
    return CASAPose(CASAPOSE_PARAMS["clade"], *args, **kwargs, learn_upsampling=False)
