"""The bare backbone entries of the reference's registry (casapose/pose_models/models/resnet.py:335-431; models_factory.py:9-13):
`Classifiers.get("resnet18")(input_shape=..., weights=None, include_top=False)` -> a model whose call returns the FIVE taps the decoders
consume, in the reference's output order (resnet.py:319): relu0, stage2_unit1_relu1, stage3_unit1_relu1, stage4_unit1_relu1, relu1 -- the
pre-activation ResNet-18 at output stride 8 (strides of stages 3, 4 replaced by dilation 2, 4; resnet.py:273-280).

Built: resnet18 with include_top=False (what `get_backbone` constructs, resnet.py:347-350).  The ImageNet classification top (global
pooling + Dense + softmax, resnet.py:307-311) and the deeper / bottleneck variants are not on the pose-estimation path: they raise
NotImplementedError by name.  The encoder kernels are the ones of the full models (casapose_amd/engine.ForwardPlan.run_encoder): no
decoder is executed.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from ...utils import h5_weights
from .model import CasaposeModel, Layer


class ResNetBackbone(CasaposeModel):
    """ResNet-18 OS-8 encoder with the Keras-like surface of CasaposeModel restricted to the encoder's layers (bn_data, conv0, bn0,
    stage<s>_unit<u>_{bn1,sc,conv1,bn2,conv2}, bn1)."""

    def __init__(self, input_shape=None, weights=None, device=None, seed=None, conv_mode=None):
        # the decoders' parameters of the underlying network are never read: the plan stops after the encoder
        super().__init__("resnet18", ver_dim=4, seg_dim=2, dims=(256, 128, 64, 32, 32), input_shape=input_shape, weights=None, device=device, seed=seed,
                         conv_mode=conv_mode)
        self._layers = self._build_layers()
        if isinstance(weights, str) and weights != "imagenet":
            self.load_weights(weights)

    def _encoder_keys(self) -> List[str]:
        return [k for k in self._params if h5_weights.is_backbone_layer(k.split(".")[0])]

    def _build_layers(self) -> List[Layer]:
        return [l for l in super()._build_layers() if h5_weights.is_backbone_layer(l.name)]

    def count_params(self) -> int:
        return int(sum(self._params[k].size for k in self._encoder_keys()))

    def get_parameters(self) -> Dict[str, np.ndarray]:
        return {k: self._params[k].copy() for k in self._encoder_keys()}

    def set_parameters(self, params: Dict[str, np.ndarray]):
        merged = dict(self._params)
        merged.update({k: v for k, v in params.items() if k in merged and h5_weights.is_backbone_layer(k.split(".")[0])})
        missing = set(self._encoder_keys()) - set(params)
        if missing:
            raise ValueError("missing parameters: %s" % sorted(missing)[:5])
        super().set_parameters(merged)

    def save_weights(self, path: str):
        """A stand-alone backbone is a top-level Keras model: its layers are top-level groups (no nested `model` group)."""
        if str(path).lower().endswith((".h5", ".hdf5", ".keras")):
            h5_weights.write_keras_h5(path, self.get_parameters(), backbone_group=None)
            return
        with open(path, "wb") as f:
            np.savez(f, **self.get_parameters())

    def training_plan(self, *a, **k):
        raise NotImplementedError("the bare backbone has no loss to train against; train it inside a casapose_* / pvnet model")

    def __call__(self, inputs, training: bool = False) -> List[torch.Tensor]:
        if training:
            raise NotImplementedError("the bare backbone runs in inference mode only (moving statistics)")
        if isinstance(inputs, (list, tuple)):
            if len(inputs) != 1:
                raise ValueError("model resnet18 expects the input `data`, got %d tensors" % len(inputs))
            inputs = inputs[0]
        img = self._to_device(inputs)
        if self.input_shape is not None and tuple(img.shape[1:]) != self.input_shape:
            raise ValueError("input `data` has shape %s, model was built for %s" % (tuple(img.shape[1:]), self.input_shape))
        b, h, w, _ = img.shape
        return [t.clone() for t in self._net.plan(b, h, w).run_encoder(img)]


def ResNet18(input_shape=None, input_tensor=None, weights=None, classes=1000, include_top=True, **kwargs):
    """resnet.py:374-383.  include_top=True (ImageNet classifier head) is not built."""
    if include_top:
        raise NotImplementedError("ResNet18(include_top=True): the ImageNet classification top (resnet.py:307-311) is not part of the pose-estimation "
                                  "path and is not built; pass include_top=False for the OS-8 encoder with its five taps")
    if input_tensor is not None:
        raise NotImplementedError("input_tensor is a Keras-graph feature without an equivalent here")
    return ResNetBackbone(input_shape=input_shape, weights=weights, device=kwargs.get("device"), seed=kwargs.get("seed"), conv_mode=kwargs.get("conv_mode"))


def _deeper(name):
    def ctor(*args, **kwargs):
        raise NotImplementedError("model `%s` is registered by the reference (resnet.py:335-343) but no model of the pose-estimation path uses it; "
                                  "only resnet18 is built for MI355X" % name)

    ctor.__name__ = name.replace("resnet", "ResNet")   # the reference's function names (resnet.py:386-431)
    return ctor


ResNet34, ResNet50, ResNet101, ResNet152 = (_deeper(n) for n in ("resnet34", "resnet50", "resnet101", "resnet152"))


def get_backbone(base_model="resnet18", input_shape=None, input_tensor=None, weights="imagenet", **kwargs):
    """resnet.py:346-371: TypeError for an undefined name."""
    table = {"resnet18": ResNet18, "resnet34": ResNet34, "resnet50": ResNet50, "resnet101": ResNet101, "resnet152": ResNet152}
    if base_model not in table:
        raise TypeError("Undefined base model type: {}".format(base_model))
    return table[base_model](input_shape=input_shape, input_tensor=input_tensor, weights=weights, include_top=False, **kwargs)
