#!/usr/bin/env python3
"""DORMANT -- cannot run in the build container (no TensorFlow, tensorflow-addons, OpenCV) and has therefore NEVER been executed; it is the recipe that closes
"parity unpinned"  on any machine that has tensorflow==2.9.1, tensorflow-addons==0.17.0 and a checkout
of the reference:

    python tools/make_tf_goldens.py --reference /path/to/casapose-checkout --out tests/golden

It IMPORTS the reference as a library (nothing of it is copied here), builds `casapose_c_gcu5` through the reference's own
`Classifiers.get(...)`, loads the seeded parameters of oracle/casapose_oracle.py into it BY VARIABLE NAME (the same
'<layer>.<field>' mapping casapose_amd/utils/h5_weights.py applies to a Keras weight file), runs the reference on seeded inputs with
training=False and stores inputs + outputs as tests/golden/tf_*.npz.  tests/test_tf_goldens.py compares the oracle (CPU) and the HIP
path (GPU) with every such file it finds; with none present it reports the parity as unpinned and skips.

What is stored (ONE run writes every file tests/test_tf_goldens.py consumes):
  tf_forward_gcu5_k5_64x96.npz   the whole forward (given mask and estimated mask)
  tf_layers_k4_12x16.npz         the decoder-2 building blocks on their own (PartialConvolution, ClassAdaptiveWeightedNormalization in
                                 inference mode, GuidedUpsampling, GuidedBilinearUpsampling, HalfSize)
  tf_voting_8obj_60x80.npz       CoordLSVotingWeighted with and without the component filter, and the RANSAC voter
                                 (ransac_voting_layer_all_masks, ransac_voting.py:276-484) with its tf.random.uniform draws REPLACED by
                                 the committed integer draws (index = draw mod tn, the convention of cp_ransac_vote_f32)
  tf_train_k5_64x64.npz          one TRAINING step's forward (training=True), the five loss values of the reference's own
                                 compute_loss (extracted from its train_casapose.py at run time) incl. keypoint_reprojection_loss
                                 through the LS voter, d loss / d output and d loss / d every trainable variable
                                 (loss_functions.py:14-344, train_casapose.py:40-145,534-592)
  tf_pnp_cases.npz               the reference's OpenCV pose recovery `pnp` (ransac_voting.py:13-57) on seeded 2-D/3-D
                                 correspondences, and BPNP_fast's forward + gradient (bpnp_layers.py:138-212,278-359)
  tf_keras_weights_k5.h5 (+.npz) the model's own `save_weights` file (h5py/Keras-written: pins casapose_amd/utils/h5_weights.py's
                                 READER) and, inside the .npz, what Keras' load_weights(by_name=True) reads back from a file written by
                                 h5_weights.write_keras_h5 (pins the WRITER)
-- the places where Appendix B of SURVEY.md had to make assumptions (tie handling in the saturated softmax, zero padding of the label
maps, the (y,x) order of the voter's result, biased batch variance, Keras' HDF5 layout, OpenCV's PnP)."""
import argparse
import ast
import importlib.util
import types
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


def reference_function(path, name, namespace):
    """Compile ONE top-level function of a reference script (train_casapose.py executes a whole training run on import, so it
    cannot be imported) into `namespace` -- the reference's own code, read where it lies at run time, nothing copied here."""
    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), namespace)
    return namespace[name]


def training_batch(rng, b, h, w, k, kp=9):
    """A consistent miniature batch: rectangular object masks, 3-D keypoints in front of a camera, identity crop; fields as in the
    reference's batch tuple (train_casapose.py:496-507)."""
    lab = np.zeros((b, h, w), np.int64)
    for n in range(b):
        for c in range(1, k):
            y0, x0 = rng.integers(0, h - h // 3), rng.integers(0, w - w // 3)
            lab[n, y0:y0 + rng.integers(h // 4, h // 2), x0:x0 + rng.integers(w // 4, w // 2)] = c
    cam = np.array([[100.0, 0, w / 2.0], [0, 100.0, h / 2.0], [0, 0, 1]], np.float32)
    p3d = rng.uniform(-20, 20, (b, k - 1, 1, kp, 3)).astype(np.float32)
    poses = np.zeros((b, k - 1, 1, 3, 4), np.float32)
    poses[..., :3, :3] = np.eye(3)
    poses[..., 2, 3] = 100.0
    camp = p3d + poses[..., None, :, 3]
    pix = camp @ cam.T
    xy = pix[..., :2] / pix[..., 2:]
    offsets = np.tile(np.array([[0.0, 0, 0, 0, 0, 0, 0, 1, w, h]], np.float32), (b, 1))
    return dict(labels=lab, target_seg=O.onehot_from_labels(lab, k, np.float32), keypoints3d=p3d, target_vert=xy[..., ::-1].copy().astype(np.float32),
                cam_mat=np.tile(cam[None], (b, 1, 1)), offsets=offsets, poses_gt=poses, img=rng.uniform(-1, 1, (b, h, w, 3)).astype(np.float32))


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

    # ---- RANSAC voter with injected draws (ransac_voting.py:276-484) -----------------------------------------
    import casapose.pose_estimation.ransac_voting as RV  # noqa: E402

    tf.config.run_functions_eagerly(True)  # the python loop must ask for its draws round by round
    hyp, objs, kpn = 128, 8, 9
    draws = np.random.default_rng(32).integers(0, 2**31 - 1, (20, 1, objs, hyp, kpn, 2), dtype=np.int64).astype(np.int32)
    labels_v = segv[0].argmax(-1)
    state = {"obj": 0, "round": 0}
    real_uniform = tf.random.uniform

    def injected_uniform(shape, minval=0, maxval=None, dtype=tf.float32, **kw):
        if dtype == tf.int32:  # the hypothesis draw of one round: [round_hyp_num, vn, 2] indices below tn
            d = draws[state["round"], 0, state["obj"]].astype(np.int64) % int(maxval)
            state["round"] += 1
            return tf.constant(d.astype(np.int32))
        return real_uniform(shape, minval=minval, maxval=maxval, dtype=dtype, **kw)  # sub-sampling above max_num: not reached here

    ransac_pts, ransac_rounds = np.zeros((1, objs, kpn, 2), np.float32), np.zeros((1, objs), np.int32)
    tf.random.uniform = injected_uniform
    try:
        for o in range(objs):
            state["obj"], state["round"] = o, 0
            m = tf.constant((labels_v == o + 1).astype(np.float32))
            ransac_pts[0, o] = RV.ransac_voting_batch(m, tf.constant(direct[0].reshape(60, 80, kpn, 2)), 0.99, 0.99, 20, 5, 30000, hyp, kpn).numpy()
            ransac_rounds[0, o] = state["round"]
    finally:
        tf.random.uniform = real_uniform
        tf.config.run_functions_eagerly(False)
    np.savez_compressed(os.path.join(args.out, "tf_voting_8obj_60x80.npz"), seg=segv, direct=direct, conf=conf, keypoints_true=kps,
                        ls=res[False], ls_filtered=res[True], ransac_draws=draws, ransac_keypoints_xy=ransac_pts, ransac_rounds=ransac_rounds,
                        tf_version=tf.__version__)

    # ---- one training step: forward(training=True), the reference's compute_loss, gradients -----------------------
    from casapose.utils.image_utils import get_all_vectorfields  # noqa: E402
    from casapose.utils.loss_functions import keypoint_reprojection_loss, proxy_voting_dist, proxy_voting_loss_v2, smooth_l1_loss  # noqa: E402

    compute_loss = reference_function(os.path.join(args.reference, "train_casapose.py"), "compute_loss",
                                      dict(tf=tf, np=np, smooth_l1_loss=smooth_l1_loss, proxy_voting_loss_v2=proxy_voting_loss_v2,
                                           proxy_voting_dist=proxy_voting_dist))
    tf.keras.backend.clear_session()
    kt, ht, wt_ = 5, 64, 64
    batch = training_batch(np.random.default_rng(7), 2, ht, wt_, kt)
    tparams = O.init_params(kt, 27, seed=5, dtype=np.float32)
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=kt, input_shape=(ht, wt_, 3), input_segmentation_shape=(ht, wt_, kt), weights=None,
                                             base_model="resnet18")
    load_params(net, tparams)
    lf = types.SimpleNamespace(mask_loss_weight=1.0, vertex_loss_weight=0.5, proxy_loss_weight=0.015, kp_loss_weight=0.007,
                               filter_vertex_with_segmentation=True, filter_high_proxy_errors=False)
    tseg, tvert = tf.constant(batch["target_seg"]), tf.constant(batch["target_vert"])
    filtered = tf.constant(batch["labels"][..., None].astype(np.int32))
    target_dirs = get_all_vectorfields(tseg, tvert, filtered, False)
    with tf.GradientTape(persistent=True) as tape:
        output_net = net([tf.constant(batch["img"]), tseg], training=True)
        tape.watch(output_net)
        output_seg, output_dirs, confidence = tf.split(output_net, [kt, 18, -1], 3)
        coords = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=kt, num_points=9, filter_estimates=False)([tseg, output_dirs, confidence])
        kp_loss, _, _ = keypoint_reprojection_loss(coords, output_seg, tf.constant(batch["poses_gt"]), tf.constant(batch["keypoints3d"]), tseg,
                                                   tf.constant(batch["cam_mat"]), tf.constant(batch["offsets"]), confidence, max_pixel_error=12.5, min_num=50,
                                                   use_bpnp_reprojection_loss=False, estimate_poses=False, confidence_regularization=True)
        loss = compute_loss(output_seg, tseg, output_dirs, target_dirs, tvert, lf, filtered, None, kp_loss=kp_loss)
    names = {l.name for l in net.layers}
    grads = {}
    for var, g in zip(net.trainable_variables, tape.gradient(loss[0], net.trainable_variables)):
        key = key_of(var.name, names)
        if key is not None and g is not None:
            grads["grad/" + key] = g.numpy()
    np.savez_compressed(os.path.join(args.out, "tf_train_k5_64x64.npz"), param_seed=5, classes=kt, output_training=output_net.numpy(), coords_yx=coords.numpy(),
                        losses=np.array([float(v) for v in loss], np.float64), dloss_doutput=tape.gradient(loss[0], output_net).numpy(),
                        target_dirs=target_dirs.numpy(), tf_version=tf.__version__, **{"batch/" + n: a for n, a in batch.items()}, **grads)
    del tape

    # ---- Keras-written HDF5 weight file, and Keras reading OUR writer's file ------------------------------------------
    kpath = os.path.join(args.out, "tf_keras_weights_k5.h5")
    net.save_weights(kpath)
    spec = importlib.util.spec_from_file_location("h5_weights", os.path.join(ROOT, "casapose_amd", "utils", "h5_weights.py"))
    h5w = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h5w)
    other = O.init_params(kt, 27, seed=99, dtype=np.float32)
    ours = os.path.join(args.out, "_ours_tmp.h5")
    h5w.write_keras_h5(ours, other)
    net.load_weights(ours, by_name=True, skip_mismatch=True)
    worst = 0.0
    for var in net.weights:
        key = key_of(var.name, names)
        if key in other:
            worst = max(worst, float(np.abs(var.numpy() - other[key]).max()))
    os.remove(ours)
    np.savez_compressed(os.path.join(args.out, "tf_keras_weights_k5.npz"), param_seed=5, classes=kt, keras_reads_our_h5_max_abs_diff=worst,
                        layer_names=np.array([l.name for l in net.layers if l.weights]), tf_version=tf.__version__)

    # ---- OpenCV pose recovery and BPnP --------------------------------------------------------------------------------
    from casapose.pose_estimation.bpnp_layers import BPNP_fast  # noqa: E402

    rng = np.random.default_rng(41)
    K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], np.float32)
    n_cases, kpn = 12, 9
    X = rng.uniform(-60, 60, (n_cases, kpn, 3)).astype(np.float32)
    Rs, ts = np.zeros((n_cases, 3, 3)), np.zeros((n_cases, 3))
    for i in range(n_cases):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w_, x_, y_, z_ = q
        Rs[i] = [[1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_)],
                 [2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_)],
                 [2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_ * x_ + y_ * y_)]]
        ts[i] = [rng.uniform(-100, 100), rng.uniform(-80, 80), rng.uniform(600, 1200)]
    camp = X @ np.transpose(Rs, (0, 2, 1)) + ts[:, None]
    pix = camp @ K.T
    x2d = (pix[..., :2] / pix[..., 2:] + rng.normal(0, 0.5, (n_cases, kpn, 2))).astype(np.float32)
    x2d[3, 2] += 60.0  # one gross outlier (RANSAC branch of solvePnPRansac, reprojectionError 12)
    x2d[7] = 0.0       # "not found": zero pose (ransac_voting.py:17-18)
    poses_cv = np.stack([RV.pnp(X[i], x2d[i], K) for i in range(n_cases)])
    pts = tf.constant(x2d[:3])
    with tf.GradientTape() as tape:
        tape.watch(pts)
        p6 = BPNP_fast(name="BPNP")([pts, tf.constant(X[:3]), tf.constant(K)])
        proj = tf.reduce_sum(p6 * tf.constant(np.arange(1, 7, dtype=np.float32)))
    np.savez_compressed(os.path.join(args.out, "tf_pnp_cases.npz"), points_3d=X, points_2d=x2d, camera=K, poses_true=np.concatenate([Rs, ts[..., None]], -1),
                        poses_cv2=poses_cv, bpnp_pose6=p6.numpy(), bpnp_upstream=np.arange(1, 7, dtype=np.float32), bpnp_grad_points=tape.gradient(proj, pts).numpy(),
                        tf_version=tf.__version__)
    print("wrote tf_forward_gcu5_k5_64x96.npz, tf_layers_k4_12x16.npz, tf_voting_8obj_60x80.npz, tf_train_k5_64x64.npz, tf_keras_weights_k5.h5/.npz, "
          "tf_pnp_cases.npz to", args.out)


if __name__ == "__main__":
    main()
