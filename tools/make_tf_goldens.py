#!/usr/bin/env python3
"""DORMANT -- cannot run in the build container (no TensorFlow) and has therefore NEVER been executed; it is the recipe that closes
"parity unpinned"  on any machine that has tensorflow==2.9.1, tensorflow-addons==0.17.0 and a checkout
of the reference:

    python tools/make_tf_goldens.py --reference /path/to/casapose-checkout --out tests/golden

It IMPORTS the reference as a library (nothing of it is copied here), builds `casapose_c_gcu5` through the reference's own
`Classifiers.get(...)`, loads the seeded parameters of oracle/casapose_oracle.py into it BY VARIABLE NAME (the same
'<layer>.<field>' mapping casapose_amd/utils/h5_weights.py applies to a Keras weight file), runs the reference on seeded inputs with
training=False and stores inputs + outputs as tests/golden/tf_*.npz.  tests/test_tf_goldens.py compares the oracle (CPU) and the HIP
path (GPU) with every such file it finds; with none present it reports the parity as unpinned and skips.

What is stored: the whole forward (given mask and estimated mask), the decoder-2 building blocks on their own (PartialConvolution,
ClassAdaptiveWeightedNormalization in inference mode, GuidedUpsampling, GuidedBilinearUpsampling, HalfSize) and CoordLSVotingWeighted
with and without the component filter -- the places where Appendix B of SURVEY.md had to make assumptions (tie handling in the
saturated softmax, zero padding of the label maps, the (y,x) order of the voter's result)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import casapose_oracle as O  # noqa: E402

FIELDS = ("kernel", "weights", "gamma", "beta", "moving_mean", "moving_variance")


def key_of(variable_name, layer_names):
    """'pv_block_6_clade/pv_block_6_clade_gamma:0' -> 'pv_block_6_clade.gamma' (same rule as keras_weights_from_h5)."""
    comps = [c for c in variable_name.split("/") if c]
    layer = next((c for c in reversed(comps[:-1]) if c in layer_names), None)
    if layer is None:
        return None
    w = comps[-1].split(":")[0]
    if w.startswith(layer + "_"):
        w = w[len(layer) + 1:]
    return layer + "." + w if w in FIELDS else None


def load_params(model, params):
    names = {l.name for l in model.layers}
    used = set()
    for v in model.weights:
        k = key_of(v.name, names)
        if k is None or k not in params:
            if "half_size" in v.name or "quater_size" in v.name or "eighth_size" in v.name:
                continue  # HalfSize keeps its identity initialisation (frozen in gcu5, pose_models.py:557-559)
            raise SystemExit("no oracle parameter for TF variable %s (mapped to %s)" % (v.name, k))
        if tuple(v.shape) != params[k].shape:
            raise SystemExit("shape mismatch for %s: TF %s, oracle %s" % (k, tuple(v.shape), params[k].shape))
        v.assign(params[k])
        used.add(k)
    missing = set(params) - used
    if missing:
        raise SystemExit("oracle parameters without a TF variable: %s" % sorted(missing)[:8])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of fraunhoferhhi/casapose (the directory holding the `casapose` package)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import tensorflow as tf  # noqa: E402
    from casapose.pose_estimation.voting_layers_2d import CoordLSVotingWeighted  # noqa: E402
    from casapose.pose_models.models._normalization_layers import (ClassAdaptiveWeightedNormalization, GuidedBilinearUpsampling,  # noqa: E402
                                                                     GuidedUpsampling, HalfSize, PartialConvolution)
    from casapose.pose_models.tfkeras import Classifiers  # noqa: E402

    tf.config.set_visible_devices([], "GPU")  # the CPU run is the reference BASELINE.json names
    os.makedirs(args.out, exist_ok=True)

    # ---- whole forward ------------------------------------------------------------------------------
    k, v, h, w = 5, 27, 64, 96
    params = O.init_params(k, v, seed=1237, dtype=np.float32)
    rng = np.random.default_rng(11)
    img = rng.uniform(-1, 1, (2, h, w, 3)).astype(np.float32)
    lab = np.zeros((2, h, w), np.int64)
    lab[:, 8:40, 10:50] = 1
    lab[:, 30:60, 40:90] = 2
    lab[0, 5:20, 60:80] = 3
    lab[1, 44:62, 4:30] = 4
    seg = O.onehot_from_labels(lab, k, np.float32)
    out = {}
    for tag, seg_shape in (("given", (h, w, k)), ("estimated", None)):
        tf.keras.backend.clear_session()
        net = Classifiers.get("casapose_c_gcu5")(ver_dim=v, seg_dim=k, input_shape=(h, w, 3), input_segmentation_shape=seg_shape, weights=None,
                                                 base_model="resnet18")
        load_params(net, params)
        out[tag] = net([img, seg] if seg_shape else [img], training=False).numpy()
    np.savez_compressed(os.path.join(args.out, "tf_forward_gcu5_k5_64x96.npz"), image=img, labels=lab.astype(np.uint8), param_seed=1237,
                        output_given_mask=out["given"], output_estimated_mask=out["estimated"], tf_version=tf.__version__)

    # ---- building blocks ------------------------------------------------------------------------------
    rng = np.random.default_rng(21)
    b, hh, ww, kk, cin, cout = 1, 12, 16, 4, 32, 32
    lab2 = np.zeros((b, hh, ww), np.int64)
    lab2[:, 2:9, 3:12] = 1
    lab2[:, 6:11, 8:15] = 2
    lab2[:, 0:3, 12:16] = 3
    mask = O.onehot_from_labels(lab2, kk, np.float32)
    x = rng.standard_normal((b, hh, ww, cin)).astype(np.float32)
    wt = (rng.standard_normal((cin, 3, 3, cout)) / 17.0).astype(np.float32)
    pc = PartialConvolution(name="pc", dim=cout, num_classes=kk)
    pc([tf.constant(x), tf.constant(mask)])  # build
    pc.set_weights([wt])
    pc_masked = pc([tf.constant(x), tf.constant(mask)]).numpy()
    pc_plain = pc([tf.constant(x)]).numpy()
    half = HalfSize(name="hs", depth=kk, trainable=False)(tf.constant(mask)).numpy()
    lo = rng.standard_normal((b, hh // 2, ww // 2, cin)).astype(np.float32)
    gu = GuidedUpsampling(name="gu")([tf.constant(lo), tf.constant(half), tf.constant(mask)]).numpy()
    gb = GuidedBilinearUpsampling(name="gb")([tf.constant(lo), tf.constant(half), tf.constant(mask)]).numpy()
    cl = ClassAdaptiveWeightedNormalization(name="cl", num_classes=kk)
    cl([tf.constant(x), tf.constant(mask)], training=False)  # build
    gamma = rng.uniform(0.5, 1.5, (kk, cin)).astype(np.float32)
    beta = (0.1 * rng.standard_normal((kk, cin))).astype(np.float32)
    mean, var = (0.1 * rng.standard_normal(cin)).astype(np.float32), rng.uniform(0.5, 1.5, cin).astype(np.float32)
    by_name = {"gamma": gamma, "beta": beta, "moving_mean": mean, "moving_variance": var}
    cl.set_weights([by_name[next(f for f in ("moving_mean", "moving_variance", "gamma", "beta") if wv.name.split(":")[0].endswith(f))] for wv in cl.weights])
    clade = cl([tf.constant(x), tf.constant(mask)], training=False).numpy()
    np.savez_compressed(os.path.join(args.out, "tf_layers_k4_12x16.npz"), labels=lab2.astype(np.uint8), x=x, weights_ihwo=wt, partial_conv=pc_masked,
                        partial_conv_one_input=pc_plain, half_size=half, low=lo, guided_up=gu, guided_bilinear_up=gb, clade_gamma=gamma, clade_beta=beta,
                        clade_mean=mean, clade_var=var, clade=clade, tf_version=tf.__version__)

    # ---- LS voting --------------------------------------------------------------------------------------
    segv, direct, conf, _labels, kps = O.synthetic_voting_inputs(1, 60, 80, num_obj=8, seed=31)
    segv, direct, conf = segv.astype(np.float32), direct.astype(np.float32), conf.astype(np.float32)
    res = {}
    for filt in (False, True):
        layer = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=9, num_points=9, filter_estimates=filt)
        res[filt] = layer([tf.constant(segv), tf.constant(direct), tf.constant(conf)]).numpy()
    np.savez_compressed(os.path.join(args.out, "tf_voting_8obj_60x80.npz"), seg=segv, direct=direct, conf=conf, keypoints_true=kps,
                        ls=res[False], ls_filtered=res[True], tf_version=tf.__version__)
    print("wrote tf_forward_gcu5_k5_64x96.npz, tf_layers_k4_12x16.npz, tf_voting_8obj_60x80.npz to", args.out)


if __name__ == "__main__":
    main()
