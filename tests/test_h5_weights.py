"""The HDF5 subset reader used for Keras weight files, pinned against a writer that emits the same structures from the
HDF5 file-format specification (no h5py / Keras file exists on this machine: parity with libhdf5 unpinned)."""
import numpy as np
import pytest

from casapose_amd.utils import h5_weights as H


def _keras_like(params, backbone="resnet18"):
    """Paths as Keras save_weights lays them out: <layer>/<layer>/<weight>:0, the backbone's layers one group deeper, custom
    layers prefixing their weight names, the CLADE's inner BatchNormalization as a sub-layer."""
    d = {}
    for k, v in params.items():
        layer, field = k.split(".")
        enc = layer.startswith(("conv0", "bn", "stage"))
        if layer.endswith("_clade") and field in ("gamma", "beta"):
            p = "%s/%s/%s_%s:0" % (layer, layer, layer, field)
        elif layer.endswith("_clade"):
            p = "%s/%s/sync_batch_normalization_7/%s:0" % (layer, layer, field)
        elif field == "weights":
            p = "%s/%s/%s_weights:0" % (layer, layer, layer)
        elif enc:
            p = "%s/%s/%s:0" % (backbone, layer, field)
        else:
            p = "%s/%s/%s:0" % (layer, layer, field)
        d[p] = v
    return d


def test_round_trip_many_groups_and_shapes(tmp_path):
    rng = np.random.default_rng(0)
    data = {"g%d/sub/w%d:0" % (i % 23, i): rng.standard_normal((i % 5 + 1, 3, i % 7 + 1)).astype(np.float32) for i in range(120)}
    data["scalar_like/v:0"] = np.array([3.5], np.float32)
    p = str(tmp_path / "t.h5")
    H.write_h5(p, data)
    assert H.is_hdf5(p)
    got = H.read_h5(p)
    assert set(got) == {"/" + k for k in data}
    for k, v in data.items():
        assert got["/" + k].dtype == np.float32 and np.array_equal(got["/" + k], v)


def test_keras_name_mapping(tmp_path):
    import casapose_oracle as O

    params = O.init_params(9, 27, seed=3, dtype=np.float32)
    p = str(tmp_path / "result_w.h5")
    H.write_h5(p, _keras_like(params))
    found = H.keras_weights_from_h5(p, {k.split(".")[0] for k in params})
    assert set(found) == set(params)
    for k in params:
        assert np.array_equal(found[k], params[k]), k


def test_unsupported_features_are_refused(tmp_path):
    p = tmp_path / "bad.h5"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes([2]) + bytes(200))       # superblock version 2
    with pytest.raises(H.H5FormatError, match="superblock version 2"):
        H.read_h5(str(p))
    q = tmp_path / "not.h5"
    q.write_bytes(b"PK\x03\x04" + bytes(64))
    assert not H.is_hdf5(str(q))
    with pytest.raises(H.H5FormatError):
        H.read_h5(str(q))


def test_keras_save_layout_round_trip(tmp_path):
    """write_keras_h5 = what `net.save_weights("result_w.h5")` writes (train_casapose.py:903): Keras' group tree (nested backbone
    model as ONE top-level layer), variable names, `layer_names` / `weight_names` / `backend` / `keras_version` attributes; the
    reader maps it back to every parameter."""
    import casapose_oracle as O

    params = O.init_params(9, 27, seed=5, dtype=np.float32)
    p = str(tmp_path / "result_w.h5")
    H.write_keras_h5(p, params)
    assert H.is_hdf5(p)
    found = H.keras_weights_from_h5(p, {k.split(".")[0] for k in params})
    assert set(found) == set(params) and all(np.array_equal(found[k], params[k]) for k in params)
    attrs = H.read_attrs(p)
    root = attrs["/"]
    assert root["backend"] == b"tensorflow" and root["keras_version"].startswith(b"2.")
    names = root["layer_names"]
    assert names.count(b"model") == 1 and b"conv0" not in names and b"pv_block_6_clade" in names and b"pv_final_conv_vertex" in names
    # every name listed in weight_names resolves to a dataset below its layer group, in Keras' variable order
    paths = set(H.read_h5(p))
    for layer in names:
        wn = attrs["/" + layer.decode()]["weight_names"]
        assert wn and all("/%s/%s" % (layer.decode(), w.decode()) in paths for w in wn)
    m = [w.decode() for w in attrs["/model"]["weight_names"]]
    # the nested model's list is trainable_weights + non_trainable_weights (Keras' _legacy_weights): every kernel / gamma / beta in layer
    # order first, every moving statistic after them -- load_weights(by_name=True) zips this list positionally
    assert m[:5] == ["bn_data/beta:0", "conv0/kernel:0", "bn0/gamma:0", "bn0/beta:0", "stage1_unit1_bn1/gamma:0"] and "stage4_unit2_conv2/kernel:0" in m
    first_moving = min(i for i, n in enumerate(m) if "moving_" in n)
    assert all("moving_" in n for n in m[first_moving:]) and not any("moving_" in n for n in m[:first_moving])
    assert m[first_moving:first_moving + 4] == ["bn_data/moving_mean:0", "bn_data/moving_variance:0", "bn0/moving_mean:0", "bn0/moving_variance:0"]
    c = [w.decode() for w in attrs["/pv_block_6_clade"]["weight_names"]]
    assert c[:2] == ["pv_block_6_clade/pv_block_6_clade_beta:0", "pv_block_6_clade/pv_block_6_clade_gamma:0"]   # add_weight order
    assert c[2] == "pv_block_6_clade/sync_batch_normalization/moving_mean:0"     # Keras' first automatic name has no suffix
    assert [w.decode() for w in attrs["/pv_block_7_clade"]["weight_names"]][3] == "pv_block_7_clade/sync_batch_normalization_1/moving_variance:0"
    assert [w.decode() for w in attrs["/pv_block_6_prepare_conv2d"]["weight_names"]] == ["pv_block_6_prepare_conv2d/pv_block_6_prepare_conv2d_weights:0"]


def test_attribute_messages_do_not_disturb_the_dataset_reader(tmp_path):
    rng = np.random.default_rng(1)
    data = {"a/b/w:0": rng.standard_normal((3, 4)).astype(np.float32), "a/c:0": rng.standard_normal(5).astype(np.float32)}
    p = str(tmp_path / "t.h5")
    H.write_h5(p, data, attrs={"": {"layer_names": [b"a"], "note": "x" * 300}, "a": {"weight_names": [b"b/w:0", b"c:0"]}, "a/b": {"k": b"v"}})
    got = H.read_h5(p)
    assert all(np.array_equal(got["/" + k], v) for k, v in data.items())
    at = H.read_attrs(p)
    assert at["/"]["layer_names"] == [b"a"] and at["/"]["note"] == b"x" * 300 and at["/a"]["weight_names"] == [b"b/w:0", b"c:0"] and at["/a/b"]["k"] == b"v"
