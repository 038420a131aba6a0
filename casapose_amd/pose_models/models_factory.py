"""Name -> constructor registry (reference: casapose/pose_models/models_factory.py:8-60).

`Classifiers.get(name)` returns a constructor with the reference's signature
(ver_dim, seg_dim, fcdim=256, s8dim=128, s4dim=64, s2dim=32, raw_dim=32, input_shape=None,
input_segmentation_shape=None, input_tensor=None, weights=None, base_model="resnet18",
backbone=None, output_lablemap=False).  Unknown names raise ValueError like the reference
(:55-56).  Registry entries whose graphs are not yet built for MI355X raise
NotImplementedError when CALLED (the name lookup still succeeds), never a silent fallback.
"""
from __future__ import annotations

import functools

from .models import casapose as _cp
from .models import pose_models as _pm
from .models import resnet as _rn


class ModelsFactory:
    _models = {
        # reference registry keys (models_factory.py:9-32)
        "resnet18": _rn.ResNet18,      # the OS-8 encoder with its five taps (include_top=False); deeper variants raise by name
        "resnet34": _rn.ResNet34,
        "resnet50": _rn.ResNet50,
        "resnet101": _rn.ResNet101,
        "resnet152": _rn.ResNet152,
        "casapose_c": _pm.CASAPoseConditional1,
        "casapose_c_gu": _pm.CASAPoseConditional2,
        "casapose_c_gcu3": _pm.CASAPoseConditional3,
        "casapose_c_gcu4": _pm.CASAPoseConditional4,
        "casapose_c_gcu5": _pm.CASAPoseConditional5,
        "pvnet_combined": _pm.PVNet,
        "casapose_custom": _cp.CASAPoseConditional,
        "casapose_c_gcu5_sw5": _pm.CASAPoseConditional6,
        "casapose_c_gcu4_sw1": _pm.CASAPoseConditional7,
        "casapose_c_gcu5_sw1": _pm.CASAPoseConditional8,
        "casapose_c_gcu4_bilat": _pm.CASAPoseConditional9,
        "casapose_c_gcu4_sw2": _pm.CASAPoseConditional10,
        "pvnet": _pm.PVNet,  # same graph with per-object (separated) vector fields: ver_dim = 2 * points * objects
    }

    @property
    def models(self):
        return self._models

    def models_names(self):
        return list(self.models.keys())

    @staticmethod
    def get_kwargs():
        # the reference injects the four Keras sub-modules here (tfkeras.py:8-14); nothing to inject
        return {}

    def inject_submodules(self, func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            merged = dict(kwargs)
            merged.update(self.get_kwargs())
            return func(*args, **merged)

        return wrapper

    def get(self, name):
        if name not in self.models_names():
            raise ValueError("No such model `{}`, available models: {}".format(name, list(self.models_names())))
        return self.inject_submodules(self.models[name])


Classifiers = ModelsFactory()
