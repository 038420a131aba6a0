"""Learning-rate schedules and the loss-weight handler of the training script.

Same constructor arguments and arithmetic as the reference's objects
(casapose/utils/learning_rate_schedules.py:6-59 ExponentialDecayLateStart, :62-115 LossWeightHandler;
tf.keras PiecewiseConstantDecay as used at train_casapose.py:334-338), evaluated on the host: a schedule is a
callable step -> float, the step being the optimizer's iteration counter BEFORE the update.
"""
from __future__ import annotations

import math
from typing import Sequence


class PiecewiseConstantDecay:
    """values[0] for step <= boundaries[0], values[i] for boundaries[i-1] < step <= boundaries[i], values[-1] after."""

    def __init__(self, boundaries: Sequence[float], values: Sequence[float], name=None):
        if len(boundaries) != len(values) - 1:
            raise ValueError("The length of boundaries should be 1 less than the length of values")
        self.boundaries, self.values, self.name = list(boundaries), list(values), name

    def __call__(self, step) -> float:
        for b, v in zip(self.boundaries, self.values):
            if step <= b:
                return float(v)
        return float(self.values[-1])

    def get_config(self):
        return {"boundaries": self.boundaries, "values": self.values, "name": self.name}


class ExponentialDecayLateStart:
    """initial * rate^p with p = 0 before decay_steps_start, then offset + (step - start)/decay_steps
    (offset 1 unless start == 0), floored when staircase."""

    def __init__(self, initial_learning_rate, decay_steps, decay_steps_start, decay_rate, staircase=False, name=None):
        self.initial_learning_rate = initial_learning_rate
        self.decay_steps = decay_steps
        self.decay_steps_start = decay_steps_start
        self.decay_rate = decay_rate
        self.staircase = staircase
        self.name = name

    def __call__(self, step) -> float:
        step = float(step)
        offset = 0.0 if self.decay_steps_start == 0 else 1.0
        p = 0.0 if step < self.decay_steps_start else offset + (step - self.decay_steps_start) / float(self.decay_steps)
        if self.staircase:
            p = math.floor(p)
        return float(self.initial_learning_rate) * float(self.decay_rate) ** p

    def get_config(self):
        return {"initial_learning_rate": self.initial_learning_rate, "decay_steps": self.decay_steps,
                "decay_steps_start": self.decay_steps_start, "decay_rate": self.decay_rate, "staircase": self.staircase,
                "name": self.name}


class LossWeightHandler:
    """Loss weights with a multiplicative per-epoch update clamped to [lo, hi] borders."""

    def __init__(self, mask_loss_weight=1.0, vertex_loss_weight=1.0, proxy_loss_weight=0.01, kp_loss_weight=1.0,
                 mask_loss_factor=1.0, vertex_loss_factor=1.0, proxy_loss_factor=1.0, kp_loss_factor=1.0,
                 mask_loss_borders=(0.0, 2.5), vertex_loss_borders=(0.000, 10.0), proxy_loss_borders=(0.000, 0.025),
                 kp_loss_borders=(0.0, 2.5), filter_vertex_with_segmentation=False, filter_high_proxy_errors=False):
        self.mask_loss_weight = mask_loss_weight
        self.vertex_loss_weight = vertex_loss_weight
        self.proxy_loss_weight = proxy_loss_weight
        self.kp_loss_weight = kp_loss_weight
        self.mask_loss_factor = mask_loss_factor
        self.vertex_loss_factor = vertex_loss_factor
        self.proxy_loss_factor = proxy_loss_factor
        self.kp_loss_factor = kp_loss_factor
        self.mask_loss_borders = mask_loss_borders
        self.vertex_loss_borders = vertex_loss_borders
        self.proxy_loss_borders = proxy_loss_borders
        self.kp_loss_borders = kp_loss_borders
        self.filter_vertex_with_segmentation = filter_vertex_with_segmentation
        self.filter_high_proxy_errors = filter_high_proxy_errors

    @staticmethod
    def clamp(n, min_max):
        return max(min_max[0], min(n, min_max[1]))

    def update(self):
        self.mask_loss_weight = self.clamp(self.mask_loss_weight * self.mask_loss_factor, self.mask_loss_borders)
        self.vertex_loss_weight = self.clamp(self.vertex_loss_weight * self.vertex_loss_factor, self.vertex_loss_borders)
        self.proxy_loss_weight = self.clamp(self.proxy_loss_weight * self.proxy_loss_factor, self.proxy_loss_borders)
        self.kp_loss_weight = self.clamp(self.kp_loss_weight * self.kp_loss_factor, self.kp_loss_borders)

    def print(self, print_fn=print):
        print_fn("==Mask loss weight: {} , vertex loss weight: {} , proxy loss weight: {} , keypoint loss weight: {}==".format(
            self.mask_loss_weight, self.vertex_loss_weight, self.proxy_loss_weight, self.kp_loss_weight))
