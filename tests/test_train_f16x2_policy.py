"""Host-side policy of the fp16-pair backward (train_engine.TrainPlan._judge_direct / _set_loss_exponent): which power of two goes on the loss, which
layers join, how the Winograd GEMMs' own exponents move with it.  Pure Python on stub objects: no GPU, no library."""
import types

import numpy as np

from casapose_amd import train_engine as TE


class _Op:
    def __init__(self, name):
        self.layer = types.SimpleNamespace(name=name)
        self.bw16 = dict(on=False, mon=1, dead=False, e=None)
        self.calls = []

    def set_direct_dgrad_f16x2(self, on, stream):
        self.bw16["on"] = on
        self.calls.append(on)

    def set_dgrad_exponent(self, entry, e, stream):
        entry["f16"]["e"] = e


def _plan(direct_names, wino_exps):
    plan = types.SimpleNamespace(loss_exp=0, f16x2_bwd_moves=[])
    ops = [_Op(n) for n in direct_names]
    slots = [(op, op.bw16, "direct") for op in ops]
    for i, e in enumerate(wino_exps):
        w = dict(f16=dict(e=e, mon=1, dead=False))
        slots.append((_Op("wino%d" % i), w["f16"], w))
    plan._bwd_slots = lambda: slots
    plan._set_bwd_exponent = TE.TrainPlan._set_bwd_exponent
    plan._set_loss_exponent = types.MethodType(TE.TrainPlan._set_loss_exponent, plan)
    plan._judge_direct = types.MethodType(TE.TrainPlan._judge_direct, plan)
    return plan, ops, slots


def test_loss_exponent_puts_the_largest_layer_at_2_pow_10_and_admits_what_fits_the_band():
    plan, ops, slots = _plan(["a", "b", "c", "d"], [5, -3])
    # unscaled maxima as a first backward would measure them: 2^-11 .. 2^-24 (a spread of 2^13)
    vals = {0: 2.0 ** -11 * 1.3, 1: 2.0 ** -15, 2: 2.0 ** -20.5, 3: 2.0 ** -24}
    plan._judge_direct(vals, 0)
    assert plan.loss_exp == 21                                   # 1.3 * 2^-11 * 2^21 = 1.3 * 2^10 in [2^10, 2^11)
    assert [op.bw16["on"] for op in ops] == [True, True, True, False]   # 2^-24 * 2^21 = 2^-3 < 1: stays on the exact split
    assert [f["e"] for _, f, e in slots if e != "direct"] == [5 - 21, -3 - 21]   # the Winograd GEMMs' own factors give the loss factor back


def test_hysteresis_and_drift():
    plan, ops, _ = _plan(["a", "b"], [])
    plan._judge_direct({0: 2.0 ** -10, 1: 2.0 ** -12}, 0)
    e0 = plan.loss_exp
    assert e0 == 20 and all(op.bw16["on"] for op in ops)
    # readings now carry the factor.  Inside [2^7, 2^13): nothing moves; a layer at 0.3 (below 1 but above 0.25) stays where it is
    plan._judge_direct({0: 2.0 ** 12.5, 1: 0.3}, 0)
    assert plan.loss_exp == e0 and ops[1].bw16["on"] and ops[1].calls == [True]
    # ... below 0.25 it leaves
    plan._judge_direct({0: 2.0 ** 12.5, 1: 0.2}, 0)
    assert not ops[1].bw16["on"]
    # the largest maximum drifts to 2^14: the exponent follows (2^14 -> [2^10, 2^11): -4); the second layer, now at 2^9 * 2^-4 = 32, joins again
    plan._judge_direct({0: 2.0 ** 14, 1: 2.0 ** 9}, 0)
    assert plan.loss_exp == e0 - 4 and ops[0].bw16["on"] and ops[1].bw16["on"]   # 2^9 * 2^-4 = 32: inside [1, 2^13] -> joins again
    # a non-finite maximum ends a layer's fp16-pair run for good
    plan._judge_direct({0: float("inf"), 1: 2.0 ** 9}, 0)
    assert ops[0].bw16["dead"] and not ops[0].bw16["on"]
    plan._judge_direct({0: 2.0 ** 10, 1: 2.0 ** 9}, 0)
    assert not ops[0].bw16["on"]


def test_all_zero_gradients_change_nothing():
    plan, ops, _ = _plan(["a"], [3])
    plan._judge_direct({0: 0.0}, 0)
    assert plan.loss_exp == 0 and not ops[0].bw16["on"]
