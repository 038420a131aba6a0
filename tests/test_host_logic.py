"""CPU tests of the host side: the registry/constructor API mirrors the reference, the
normalisation folding is exact, the config parser keeps the reference's precedence, and the
product path refuses to run without a GPU instead of falling back."""
import numpy as np
import pytest
import torch

import casapose_oracle as O


def test_registry_names_match_reference_and_unknown_name_raises():
    from casapose_amd.pose_models import model_factory, tfkeras
    from casapose_amd.pose_models.models_factory import Classifiers

    assert tfkeras.Classifiers is Classifiers and model_factory.Classifiers is Classifiers
    names = Classifiers.models_names()
    for n in ("resnet18", "casapose_c", "casapose_c_gu", "casapose_c_gcu3", "casapose_c_gcu4", "casapose_c_gcu5", "pvnet_combined",
              "casapose_custom", "casapose_c_gcu5_sw5", "casapose_c_gcu4_sw1", "casapose_c_gcu5_sw1", "casapose_c_gcu4_bilat",
              "casapose_c_gcu4_sw2", "pvnet"):
        assert n in names
    assert len(names) == 18  # models_factory.py:9-32
    with pytest.raises(ValueError, match="No such model"):
        Classifiers.get("does_not_exist")
    with pytest.raises(NotImplementedError, match="resnet34"):   # registered by the reference; no model of the path uses the deeper backbones
        Classifiers.get("resnet34")(input_shape=(64, 64, 3))
    with pytest.raises(NotImplementedError, match="include_top"):  # resnet18 itself is built (GPU tests) -- without the ImageNet classifier top
        Classifiers.get("resnet18")(input_shape=(224, 224, 3))
    from casapose_amd.pose_models.models.resnet import get_backbone

    with pytest.raises(TypeError, match="Undefined base model type"):   # resnet.py:368-369
        get_backbone("vgg16")
    from casapose_amd._lib import CasaposeHipError

    with pytest.raises(CasaposeHipError):      # pvnet with separated vector fields (9 + 18*8 output channels) constructs -- on a GPU (no CPU fallback)
        Classifiers.get("pvnet")(ver_dim=144, seg_dim=9, device="cpu")


def test_no_cpu_fallback():
    from casapose_amd import _lib
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from casapose_amd.pose_models.tfkeras import Classifiers

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.CasaposeHipError, match="no CPU fallback"):
        Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=9, input_shape=(64, 96, 3))
    t = torch.zeros(1, 8, 8, 9)
    with pytest.raises(_lib.CasaposeHipError, match="no CPU fallback"):
        CoordLSVotingWeighted("v", 9)([t, torch.zeros(1, 8, 8, 18), t])


def test_decoder_params_surface():
    from casapose_amd.pose_models.models.casapose import CASAPOSE_PARAMS, DecoderParams

    assert DecoderParams._fields == ("weighted_clade", "partial_conv", "guided_upsampling", "bilinear_upsampling", "reuse_conv")
    p = CASAPOSE_PARAMS["clade"]
    assert len(p) == 5 and [x.guided_upsampling for x in p] == [False, True, True, True, False]
    assert all(x.weighted_clade and x.partial_conv and not x.bilinear_upsampling and not x.reuse_conv for x in p)


def test_bn_and_clade_folding_match_the_oracle():
    from casapose_amd.engine import fold_bn, fold_clade

    rng = np.random.default_rng(0)
    c, k = 16, 4
    p = {"n.gamma": rng.uniform(0.5, 1.5, c), "n.beta": rng.standard_normal(c), "n.moving_mean": rng.standard_normal(c),
         "n.moving_variance": rng.uniform(0.5, 1.5, c)}
    x = rng.standard_normal((2, 3, 5, c))
    s, b = fold_bn(p, "n")
    assert np.allclose(x * s + b, O.batchnorm_inference(x, p["n.gamma"], p["n.beta"], p["n.moving_mean"], p["n.moving_variance"]), atol=1e-5)
    q = {"d.beta": p["n.beta"][:3], "d.moving_mean": p["n.moving_mean"][:3], "d.moving_variance": p["n.moving_variance"][:3]}
    s, b = fold_bn(q, "d", pad_to=4)  # bn_data: no gamma, padded to the 4-channel operand
    assert s.shape == (4,) and s[3] == 0 and b[3] == 0
    assert np.allclose(x[..., :3] * s[:3] + b[:3], O.batchnorm_inference(x[..., :3], None, q["d.beta"], q["d.moving_mean"], q["d.moving_variance"]), atol=1e-5)
    pc = {"c.gamma": rng.uniform(0.5, 1.5, (k, c)), "c.beta": rng.standard_normal((k, c)), "c.moving_mean": p["n.moving_mean"],
          "c.moving_variance": p["n.moving_variance"]}
    lab = rng.integers(0, k, (2, 3, 5))
    ts, tb = fold_clade(pc, "c")
    ref = O.clade_weighted(x, O.onehot_from_labels(lab, k), pc["c.gamma"], pc["c.beta"], pc["c.moving_mean"], pc["c.moving_variance"])
    assert np.allclose(x * ts[lab] + tb[lab], ref, atol=1e-5)


def test_initial_parameters_cover_the_oracle_parameter_set():
    from casapose_amd.pose_models.models.model import initial_parameters

    mine = initial_parameters(9, 27, (256, 128, 64, 32, 32), seed=0)
    ref = O.init_params(9, 27)
    assert set(mine) == set(ref)
    for k in mine:
        assert mine[k].shape == ref[k].shape, k
    n = sum(v.size for k, v in mine.items() if k.endswith((".kernel", ".weights")))
    assert n == 11171008 + 3560256  # encoder + decoder conv weights (K=9, ver_dim=27)
    w = mine["pv_block_6_prepare_conv2d.weights"]
    assert w.shape == (512, 3, 3, 256) and abs(w).max() <= np.sqrt(6.0 / (9 * 512))  # he_uniform, fan_in 9*Cin


@pytest.mark.parametrize("variant", sorted(O.VARIANTS))
def test_variant_parameter_sets_and_oracle_sharing(variant):
    """every registry variant: the model's initial parameter set equals the oracle's (names + shapes); for the shared-weight models the
    oracle's forward must really use ONE weight set in both decoders: perturbing it changes the segmentation logits (decoder 1) and
    the vector field (decoder 2), and with reuse_first block 6 reads block 1's convolution output."""
    from casapose_amd.pose_models.models.model import initial_parameters

    part, _ = O.VARIANTS[variant]
    sharing = O.SHARED.get(variant, {})
    mine = initial_parameters(4, 27, (256, 128, 64, 32, 32), seed=0, partial=part, **sharing)
    ref = O.init_params(4, 27, partial=part, dtype=np.float64, **sharing)
    assert set(mine) == set(ref)
    assert all(mine[k].shape == ref[k].shape for k in mine)
    if not sharing:
        return
    rng = np.random.default_rng(3)
    img = rng.uniform(-1, 1, (1, 16, 24, 3))
    lab = np.zeros((1, 16, 24), np.int64)
    lab[:, 4:12, 6:18] = 1
    seg = O.onehot_from_labels(lab, 4, np.float64)
    base = O.casapose_c_gcu5(ref, img, seg_input=seg, variant=variant)
    i = max(j for j in range(5) if sharing["shared"][j])
    bumped = dict(ref)
    bumped[O.shared_key(i)] = ref[O.shared_key(i)] * 1.5
    out = O.casapose_c_gcu5(bumped, img, seg_input=seg, variant=variant)
    assert np.abs(out[..., :4] - base[..., :4]).max() > 1e-6 and np.abs(out[..., 4:] - base[..., 4:]).max() > 1e-6
    assert ("pv_block_6_prepare_conv2d.weights" in ref) == (not sharing["reuse_first"] and not sharing["shared"][0])


def test_config_precedence_and_postprocessing(tmp_path):
    from casapose_amd.utils.config_parser import parse_config

    opt = parse_config([])
    assert opt.modelname == "casapose_cond_weighted" and opt.imagesize == (448, 448) and opt.batchsize == 32
    assert opt.outf == "output/tmp" and opt.evalf == "output/tmp/tmp" and 1 <= opt.manualseed < 10000
    ini = tmp_path / "c.ini"
    ini.write_text("[defaults]\nmodelname: casapose_c_gcu5\nimagesize_test: 480, 640\nbatchsize: 4\nestimate_coords: 1\n"
                   "lr_epochs_steps: 50,75,90\ngpuids: 0,1\nmanualseed: 1237\nobject: obj_000001,obj_000005\n")
    opt = parse_config(["-c", str(ini)])
    assert opt.modelname == "casapose_c_gcu5" and opt.imagesize_test == (480, 640) and opt.batchsize == 4
    assert opt.estimate_coords is True and opt.lr_epochs_steps == [50, 75, 90] and opt.gpuids == [0, 1] and opt.manualseed == 1237
    opt = parse_config(["-c", str(ini), "--batchsize", "16", "--estimate_coords", "no", "--gpuids", "-1"])
    assert opt.batchsize == 16 and opt.estimate_coords is False and opt.gpuids == [-1]  # command line wins
    assert opt.objects_to_copy.tolist() == [[0, 0]] and opt.objects_in_input_network == 0
    with pytest.raises(SystemExit):
        parse_config(["--estimate_coords", "maybe"])  # argparse.ArgumentTypeError -> exit, like the reference


def test_shipped_configs_parse():
    import os

    from casapose_amd.utils.config_parser import parse_config

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    o8 = parse_config(["-c", os.path.join(root, "config", "config_8.ini")])
    o13 = parse_config(["-c", os.path.join(root, "config", "config_13.ini")])
    assert o8.modelname == o13.modelname == "casapose_c_gcu5"
    assert len(o8.object.split(",")) == 8 and len(o13.object.split(",")) == 13
    assert o8.imagesize == (448, 448) and o8.imagesize_test == (480, 640) and o8.train_vectors_with_ground_truth is True
    assert o8.proxy_loss_weight == 0.015 and o8.keypoint_loss_weight == 0.007 and o8.max_keypoint_pixel_error == 12.5
    assert o13.filter_test_with_gt is True and o8.filter_test_with_gt is False


def test_voting_record_detection():
    from casapose_amd.pose_estimation.voting_layers_2d import _as_record

    out = torch.arange(2 * 4 * 5 * 36, dtype=torch.float32).reshape(2, 4, 5, 36)
    s, d, c = torch.split(out, [9, 18, 9], dim=3)
    rec, offs = _as_record(s, d, c)
    assert rec.data_ptr() == out.data_ptr() and offs == [0, 9, 27]  # zero-copy
    rec2, offs2 = _as_record(s.contiguous(), d.contiguous(), c.contiguous())
    assert rec2.shape[3] == 36 and offs2 == [0, 9, 27] and torch.equal(rec2, out)


# ---- training-side host logic ---------------------------------------------------------------------
def test_learning_rate_schedules_and_loss_weights():
    from casapose_amd.utils.learning_rate_schedules import ExponentialDecayLateStart, LossWeightHandler, PiecewiseConstantDecay

    # train_casapose.py:334-337: boundaries = epochs*batches - 1, values = lr * decay^i
    batches = 10
    boundaries = (np.array([5, 7]) * batches - 1).tolist()
    values = (np.power(0.5, np.arange(3)) * 1e-3).tolist()
    s = PiecewiseConstantDecay(boundaries, values)
    assert [s(i) for i in (0, 49, 50, 69, 70, 1000)] == [1e-3, 1e-3, 5e-4, 5e-4, 2.5e-4, 2.5e-4]
    with pytest.raises(ValueError):
        PiecewiseConstantDecay([1, 2], [1.0, 2.0])
    e = ExponentialDecayLateStart(1e-3, decay_steps=10, decay_steps_start=20, decay_rate=0.5, staircase=True)
    assert [e(i) for i in (0, 19, 20, 29, 30)] == [1e-3, 1e-3, 5e-4, 5e-4, 2.5e-4]
    e0 = ExponentialDecayLateStart(1e-3, decay_steps=10, decay_steps_start=0, decay_rate=0.5, staircase=False)
    assert abs(e0(5) - 1e-3 * 0.5 ** 0.5) < 1e-12
    h = LossWeightHandler(mask_loss_weight=1.0, vertex_loss_weight=0.5, proxy_loss_weight=0.015, kp_loss_weight=0.007, proxy_loss_factor=2.0,
                          kp_loss_factor=0.5)
    h.update()
    assert (h.mask_loss_weight, h.vertex_loss_weight, h.proxy_loss_weight, h.kp_loss_weight) == (1.0, 0.5, 0.025, 0.0035)
    # the reference's positional order (weights, factors, borders, flags: learning_rate_schedules.py:63-79), attribute surface, clamping
    p = LossWeightHandler(1.0, 0.5, 0.015, 0.007, 3.0, 1.0, 1.0, 0.1, (0.0, 2.5), (0.0, 10.0), (0.0, 0.025), (0.005, 2.5), True)
    assert p.filter_vertex_with_segmentation is True and p.filter_high_proxy_errors is False and p.kp_loss_borders == (0.005, 2.5)
    p.update()
    assert (p.mask_loss_weight, p.kp_loss_weight) == (2.5, 0.005)          # clamped at the upper / lower border
    p.vertex_loss_weight = 4.0
    p.vertex_loss_factor = 3.0
    p.update()
    assert p.vertex_loss_weight == 10.0 and p.clamp(7, (0, 5)) == 5
    lines = []
    p.print(lines.append)
    assert lines == ["==Mask loss weight: 2.5 , vertex loss weight: 10.0 , proxy loss weight: 0.015 , keypoint loss weight: 0.005=="]
    d = LossWeightHandler()
    assert (d.mask_loss_weight, d.vertex_loss_weight, d.proxy_loss_weight, d.kp_loss_weight, d.proxy_loss_borders) == (1.0, 1.0, 0.01, 1.0, (0.0, 0.025))
    with pytest.raises(TypeError):
        LossWeightHandler(colour_loss_weight=1.0)
    with pytest.raises(AttributeError):
        d.colour_loss_weight


def test_crop_affine_and_projection_match_oracle():
    import torch_train_ref as R
    from casapose_amd.train_engine import crop_to_image_affine, project_keypoints

    rng = np.random.default_rng(0)
    b = 5
    offsets = np.stack([rng.uniform(0, 30, b), rng.uniform(0, 60, b), np.zeros(b), np.zeros(b), rng.uniform(-5, 5, b), rng.uniform(-5, 5, b),
                        rng.uniform(-30, 30, b), rng.uniform(0.7, 1.3, b), np.full(b, 640.0), np.full(b, 480.0)], 1)
    A = crop_to_image_affine(offsets).reshape(b, 2, 3)
    assert np.allclose(A, R.crop_to_image_affine(offsets), rtol=1e-6, atol=1e-4)
    # no augmentation: crop pixel + crop origin
    ident = np.array([[16.0, 96.0, 0, 0, 0, 0, 0, 1, 640, 480]])
    A0 = crop_to_image_affine(ident).reshape(2, 3)
    assert np.allclose(A0, [[1, 0, 96], [0, 1, 16]], atol=1e-5)
    K = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1]])
    xyz = rng.uniform(-50, 50, (b, 9, 3))
    RT = np.concatenate([np.tile(np.eye(3), (b, 1, 1)), np.tile(np.array([[0.0], [0.0], [800.0]]), (b, 1, 1))], axis=2)
    got = project_keypoints(xyz, K, RT)
    assert np.allclose(got, R.project_points(xyz, K, RT), rtol=1e-5, atol=1e-3)


def test_flat_parameter_buffer_follows_forward_order_for_gradient_buckets():
    """the backward completes the flat gradient from the end towards the start, so each bucket of consecutive layers is a contiguous
    slice that is final when the backward has passed the bucket's first layer (train_engine.TrainPlan.backward)."""
    from casapose_amd.train_engine import BUCKET_STARTS, ParamStore, forward_order

    params = O.init_params(4, 27, seed=1, dtype=np.float32)
    st = ParamStore(dict(reversed(list(params.items()))), torch.device("cpu"))  # input order must not matter
    rank = {n: i for i, n in enumerate(forward_order())}
    names = sorted(st.offsets, key=lambda n: st.offsets[n][0])
    assert all(n.split(".")[0] in rank for n in names), "every trainable layer has a place in the forward order"
    ranks = [rank[n.split(".")[0]] for n in names]
    assert ranks == sorted(ranks)
    assert [rank[b] for b in BUCKET_STARTS] == sorted(rank[b] for b in BUCKET_STARTS) and rank[BUCKET_STARTS[0]] == 0
    assert all(st.offsets[n][0] % 4 == 0 for n in names)
    np.testing.assert_array_equal(st.view("conv0.kernel").numpy(), params["conv0.kernel"])


def test_casapose_alias_package_serves_the_reference_import_surface():
    """`from casapose... import ...` as the reference's scripts write it (tfkeras.py:17; train_casapose.py:12-31; test_casapose.py:10-21)
    resolves to the SAME module objects as casapose_amd (one copy of every module, shared library handle and caches)."""
    import importlib

    import casapose
    import casapose_amd

    pairs = ["pose_models.tfkeras", "pose_models.models_factory", "pose_models.model_factory", "pose_estimation.voting_layers_2d", "pose_estimation.ransac_voting",
             "pose_estimation.pose_evaluation", "utils.config_parser", "utils.io_utils", "utils.learning_rate_schedules", "data_handler.vectorfield_dataset",
             "utils.loss_functions", "utils.image_utils", "pose_models.models.resnet"]
    for name in pairs:
        mod = importlib.import_module("casapose." + name)
        assert mod is importlib.import_module("casapose_amd." + name), name
        # the alias import must leave the real module's spec in place (round-2 advisor finding: reload() was a no-op, relative imports warned)
        assert mod.__spec__.name == "casapose_amd." + name and mod.__package__ == mod.__spec__.parent, (name, mod.__spec__)
    lrs = importlib.import_module("casapose.utils.learning_rate_schedules")
    assert importlib.reload(lrs) is lrs and hasattr(lrs, "LossWeightHandler")
    from casapose.pose_models.tfkeras import Classifiers
    from casapose_amd.pose_models.tfkeras import Classifiers as C2

    assert Classifiers is C2 and "casapose_c_gcu5" in Classifiers.models_names()
    import pytest

    with pytest.raises(ModuleNotFoundError, match="casapose_amd.utils.draw_utils"):
        importlib.import_module("casapose.utils.draw_utils")
    assert casapose.__path__ == [] and casapose_amd.__name__ == "casapose_amd"


def test_split_kernel_host_packers(hip_lib):
    """cp_conv_pack_weights_split_host / cp_conv_pack_head_split_host (pure host code): the fp32 image of the bf16-pipe kernel's fragment
    stream -- [pass][step][cout block][64 lanes][8 k] with 9 steps per 16-channel slice and 3 tap-major steps for the 4-channel image
    source; passes of 64 output channels above 64."""
    import ctypes as C

    rng = np.random.default_rng(0)
    # 32 + image -> 32 (decoder block 5 / 10)
    w = rng.standard_normal((3, 3, 35, 32)).astype(np.float32)
    ch, re = (C.c_int * 2)(32, 4), (C.c_int * 2)(32, 3)
    n = hip_lib.cp_conv_split_weight_floats(32, 2, ch)
    assert n == (2 * 9 + 3) * 1 * 512 and hip_lib.cp_conv_split_weight_bytes(32, 2, ch, 3) == (2 * 9 + 3) * 3 * 1024
    dst = np.full(n, np.nan, np.float32)
    assert hip_lib.cp_conv_pack_weights_split_host(w.ctypes.data, 0, 32, 2, ch, re, dst.ctypes.data) == 0
    f = dst.reshape(21, 64, 8)
    for c, t, l, e in ((0, 0, 0, 0), (1, 4, 37, 5), (1, 8, 63, 7)):      # slice c, tap t: W[co = l & 31][channel 16c + 8(l >> 5) + e]
        assert f[c * 9 + t, l, e] == w[t // 3, t % 3, 16 * c + 8 * (l >> 5) + e, l & 31]
    for s3, l, e in ((0, 3, 0), (1, 40, 6), (2, 5, 2)):                   # image step s3: tap 4 s3 + 2 (l >> 5) + (e >> 2), channel e & 3
        t, cch = 4 * s3 + 2 * (l >> 5) + (e >> 2), e & 3
        want = w[t // 3, t % 3, 32 + cch, l & 31] if (t < 9 and cch < 3) else 0.0
        assert f[18 + s3, l, e] == want
    assert f[20, 40, 0] == 0.0 and not np.isnan(dst).any()               # tap 10 does not exist
    # 32 -> 160 in the partial-convolution layout [Cin,3,3,Cout]: three passes of 64 output channels, zero rows beyond 160
    w2 = rng.standard_normal((32, 3, 3, 160)).astype(np.float32)
    ch1, re1 = (C.c_int * 2)(32, 0), (C.c_int * 2)(32, 0)
    n2 = hip_lib.cp_conv_split_weight_floats(160, 1, ch1)
    assert n2 == 3 * 18 * 2 * 512
    d2 = np.empty(n2, np.float32)
    assert hip_lib.cp_conv_pack_weights_split_host(w2.ctypes.data, 1, 160, 1, ch1, re1, d2.ctypes.data) == 0
    g = d2.reshape(3, 18, 2, 64, 8)
    assert g[1, 9 + 5, 1, 33, 2] == w2[16 + 8 + 2, 5 // 3, 5 % 3, 64 + 32 + 1]
    assert g[2, 0, 0, 7, 0] == w2[0, 0, 0, 128 + 7] and (g[2, :, 1] == 0).all()
    # fused head: step m, lane (q, kh), element e <- Wh[8 (2m + (e >> 2)) + 4 kh + (e & 3)][q]
    wh = rng.standard_normal((32, 9)).astype(np.float32)
    dh = np.empty(1024, np.float32)
    assert hip_lib.cp_conv_pack_head_split_host(wh.ctypes.data, 9, dh.ctypes.data) == 0
    hh = dh.reshape(2, 64, 8)
    assert hh[1, 32 + 4, 6] == wh[8 * (2 + 1) + 4 + 2, 4] and hh[0, 20, 3] == 0.0      # q = 20 >= 9: zero


def _unjson(v):
    if isinstance(v, dict):
        if "__ndarray__" in v:
            return np.asarray(v["__ndarray__"], dtype=v["dtype"])
        if "__tuple__" in v:
            return tuple(_unjson(e) for e in v["__tuple__"])
        if "__list__" in v:
            return [_unjson(e) for e in v["__list__"]]
    return v


def test_config_parser_matches_the_reference_parser(tmp_path, monkeypatch):
    """tests/golden/config_parser_ref.json holds what the REFERENCE's own parse_config() (casapose/utils/config_parser.py:7-170, executed in
    this container by tests/golden/make_config_golden.py -- it needs no TensorFlow) returns for its config_8.ini / config_13.ini under six
    command lines: every option must come out with the same name, type and value here."""
    import json
    import os

    from casapose_amd.utils.config_parser import parse_config

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(root, "tests", "golden", "config_parser_ref.json")))
    monkeypatch.chdir(root)   # `objects_to_copy_list: config/objects_to_copy.csv` is relative to the repository root, as in the reference
    assert len(cases) == 6
    for n, case in enumerate(cases):
        argv = list(case["argv"])
        if case["ini"] is not None:
            ini = tmp_path / ("case%d.ini" % n)
            ini.write_text("".join("[%s]\n%s\n" % (sec, "\n".join("%s: %s" % kv for kv in items.items())) for sec, items in case["ini"].items()))
            argv[argv.index("-c") + 1] = str(ini)
        opt = vars(parse_config(argv))
        want = {k: _unjson(v) for k, v in case["opt"].items()}
        assert set(opt) == set(want), (n, set(opt) ^ set(want))
        for k, w in want.items():
            g = opt[k]
            if k == "manualseed" and "manualseed" not in (case["ini"] or {}).get("defaults", {}):
                assert isinstance(g, int) and 1 <= g <= 10000   # drawn at random when absent
                continue
            assert type(g) is type(w), (n, k, type(g), type(w))
            if isinstance(w, np.ndarray):
                assert g.dtype == w.dtype and g.shape == w.shape and (g == w).all(), (n, k)
            else:
                assert g == w, (n, k, g, w)
    # the two config files this repository ships carry the same options as the reference's own files
    for n, name in ((0, "config/config_8.ini"), (1, "config/config_13.ini")):
        opt = vars(parse_config(["-c", name]))
        for k, w in cases[n]["opt"].items():
            w = _unjson(w)
            assert (opt[k] == w).all() if isinstance(w, np.ndarray) else opt[k] == w, (name, k, opt[k], w)


def test_loss_weight_handler_equals_the_reference_class():
    """`LossWeightHandler.__init__` / `clamp` / `update` of the reference (learning_rate_schedules.py:62-109) are plain Python: executed by
    tests/golden/make_geometry_golden.py (extracted with `ast`), their weight traces over several update() calls -- defaults, config_8.ini's
    weights passed positionally, growing / shrinking factors hitting both borders, all fourteen positional arguments -- pin this
    repository's table-driven restatement (same constructor signature, same attributes)."""
    import json
    import os

    from casapose_amd.utils.learning_rate_schedules import LossWeightHandler

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geometry_ref.json")))["loss_weight_handler"]
    assert len(ref) == 4
    for c in ref:
        args = [tuple(a) if isinstance(a, list) else a for a in c["args"]]
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c["kw"].items()}
        h = LossWeightHandler(*args, **kw)
        for row in c["trace"]:
            assert [h.mask_loss_weight, h.vertex_loss_weight, h.proxy_loss_weight, h.kp_loss_weight] == row
            h.update()
        assert [h.filter_vertex_with_segmentation, h.filter_high_proxy_errors] == c["flags"]


def test_residual_unit_layer_names_equal_the_reference_helper():
    """`handle_block_names` (resnet.py:20-26, executed by the golden generator) gives the stems the reference appends '1' / '2' to (resnet.py:78-103:
    bn_name + '1', conv_name + '1', bn_name + '2', conv_name + '2', sc_name): the Keras layer names -- and with them the variable names of a
    weight file -- that the oracle's parameter dictionary, the HDF5 writer's backbone order and the engines use."""
    import json
    import os

    import casapose_oracle as O
    from casapose_amd.utils import h5_weights as H

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geometry_ref.json")))["block_names"]
    convs = {n for n, _, _, _ in O.encoder_conv_specs()}
    bns = {n for n, _, _ in O.encoder_bn_specs()}
    order = H.keras_backbone_layer_order()
    for stage, block, conv_name, bn_name, relu_name, sc_name in ref:
        assert conv_name + "1" in convs and conv_name + "2" in convs and bn_name + "1" in bns and bn_name + "2" in bns
        assert (sc_name in convs) == (block == 0)                      # the 1x1 shortcut exists in the first unit of a stage only (cut = 'post')
        i = order.index(bn_name + "1")
        assert order[i:i + 4] == [bn_name + "1", conv_name + "1", bn_name + "2", conv_name + "2"]
    assert len(convs) == 1 + 4 * (2 * 2 + 1) and len(ref) == 8


def test_model_registry_equals_the_reference_registry():
    """The keys of `ModelsFactory._models` and the class each key names (pose_models/models_factory.py:9-35, read off the dict literal's syntax
    tree by the golden generator) against this repository's registry: same 18 names in the same order, each bound to a callable of the same
    name (`Classifiers.get(name)` is the drop-in entry the reference's scripts use, tfkeras.py:6-17)."""
    import json
    import os

    from casapose_amd.pose_models.models_factory import ModelsFactory

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geometry_ref.json")))["registry"]
    mine = ModelsFactory()
    assert mine.models_names() == [k for k, _ in ref] and len(ref) == 18
    for key, cls_name in ref:
        assert mine.models[key].__name__ == cls_name, (key, mine.models[key].__name__, cls_name)
    with pytest.raises(ValueError, match="No such model"):
        mine.get("casapose_does_not_exist")
