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
    """The four loss weights of compute_loss (mask, vertex, proxy, kp) with a per-epoch multiplicative schedule: update() multiplies
    each weight by its factor and clips it into its [low, high] border pair.  Interface of the reference's object
    (learning_rate_schedules.py:62-115: positional order and keyword names of the constructor, the attributes `<term>_loss_weight`,
    `<term>_loss_factor`, `<term>_loss_borders`, the two filter flags read by compute_loss, update(), clamp(), print()); the state lives in
    one table keyed by term instead of fourteen hand-written attributes."""

    TERMS = ("mask", "vertex", "proxy", "kp")
    _DEFAULT_WEIGHT = {"mask": 1.0, "vertex": 1.0, "proxy": 0.01, "kp": 1.0}
    _DEFAULT_BORDERS = {"mask": (0.0, 2.5), "vertex": (0.0, 10.0), "proxy": (0.0, 0.025), "kp": (0.0, 2.5)}
    _FIELDS = ("weight", "factor", "borders")

    def __init__(self, *args, filter_vertex_with_segmentation=False, filter_high_proxy_errors=False, **kw):
        # positional order of the reference: 4 weights, 4 factors, 4 border pairs, then the two flags
        names = ["%s_loss_%s" % (t, f) for f in self._FIELDS for t in self.TERMS] + ["filter_vertex_with_segmentation", "filter_high_proxy_errors"]
        if len(args) > len(names):
            raise TypeError("LossWeightHandler takes at most %d positional arguments (%d given)" % (len(names), len(args)))
        for n, v in zip(names, args):
            if n in kw:
                raise TypeError("LossWeightHandler got multiple values for argument %r" % n)
            kw[n] = v
        self.filter_vertex_with_segmentation = kw.pop("filter_vertex_with_segmentation", filter_vertex_with_segmentation)
        self.filter_high_proxy_errors = kw.pop("filter_high_proxy_errors", filter_high_proxy_errors)
        self._state = {}
        for t in self.TERMS:
            self._state[t] = {"weight": kw.pop(t + "_loss_weight", self._DEFAULT_WEIGHT[t]), "factor": kw.pop(t + "_loss_factor", 1.0),
                              "borders": kw.pop(t + "_loss_borders", self._DEFAULT_BORDERS[t])}
        if kw:
            raise TypeError("LossWeightHandler got an unexpected keyword argument %r" % sorted(kw)[0])

    # `<term>_loss_<field>` attributes resolve into the table (read and write)
    @classmethod
    def _split(cls, name):
        for t in cls.TERMS:
            for f in cls._FIELDS:
                if name == "%s_loss_%s" % (t, f):
                    return t, f
        return None

    def __getattr__(self, name):
        key = None if name.startswith("_") else self._split(name)
        if key is None:
            raise AttributeError(name)
        return self._state[key[0]][key[1]]

    def __setattr__(self, name, value):
        key = None if name.startswith("_") else self._split(name)
        if key is None:
            object.__setattr__(self, name, value)
        else:
            self._state[key[0]][key[1]] = value

    @staticmethod
    def clamp(n, min_max):
        lo, hi = min_max
        return max(lo, min(n, hi))

    def update(self):
        for st in self._state.values():
            st["weight"] = self.clamp(st["weight"] * st["factor"], st["borders"])

    def print(self, print_fn=print):
        w = [self._state[t]["weight"] for t in self.TERMS]
        print_fn("==Mask loss weight: {} , vertex loss weight: {} , proxy loss weight: {} , keypoint loss weight: {}==".format(*w))
