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


# --------------------------------------------------------------------------------------------------
# against the real HDF5 library (h5py 3.3.0 / libhdf5 1.10.6 of the image's conda environment -- what Keras itself reads and writes with)
# --------------------------------------------------------------------------------------------------
import importlib.util  # noqa: E402
import os  # noqa: E402
import subprocess  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONDA_PY = "/opt/conda/bin/python3.9"


def _maker():
    spec = importlib.util.spec_from_file_location("make_h5py_golden", os.path.join(GOLDEN, "make_h5py_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _have_h5py():
    return os.path.exists(CONDA_PY) and subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True).returncode == 0


def test_reader_parses_a_file_written_by_h5py():
    """tests/golden/h5py_keras_layout.h5 was written by h5py (make_h5py_golden.py write) in Keras' save_weights layout: nested backbone
    group, CLADE layer with its inner normalisation scope, PartialConvolution, fixed-length `layer_names` / `weight_names` arrays and the
    VARIABLE-length scalar strings h5py makes of `backend` / `keras_version`.  Every dataset must come back bit-equal under its
    '<layer>.<field>' key -- the first bytes of libhdf5 this reader has met (rounds 1-2: reader and writer had only met each other)."""
    mk = _maker()
    want = mk.params(20)
    path = os.path.join(GOLDEN, "h5py_keras_layout.h5")
    assert H.is_hdf5(path)
    found = H.keras_weights_from_h5(path, {k.split(".")[0] for k in want})
    assert set(found) == set(want) and all(found[k].dtype == np.float32 and np.array_equal(found[k], want[k]) for k in want)
    attrs = H.read_attrs(path)
    assert attrs["/"]["backend"] == b"tensorflow" and attrs["/"]["keras_version"] == b"2.9.0"
    assert attrs["/"]["layer_names"] == [b"model", b"pv_block_10_prepare_conv2d", b"pv_block_10_clade", b"pv_block_5_bn", b"pv_final_conv_vertex"]
    assert attrs["/model"]["weight_names"][:3] == [b"bn_data/beta:0", b"conv0/kernel:0", b"bn0/gamma:0"]
    assert attrs["/pv_block_10_clade"]["weight_names"][2] == b"pv_block_10_clade/sync_batch_normalization/moving_mean:0"


@pytest.mark.skipif(not _have_h5py(), reason="needs the image's conda interpreter with h5py")
def test_h5py_reads_files_written_by_our_writer(tmp_path):
    """The other direction: libhdf5 (through h5py, read the way Keras' loader reads: f.attrs['layer_names'], g.attrs['weight_names'],
    np.asarray(g[name])) must accept write_keras_h5's output -- the reduced layer set and the WHOLE 14.75 M-parameter network.  This
    comparison found two writer bugs in round 3 (local-heap free-list sentinel, group B-tree nodes shorter than their fixed size)."""
    import casapose_oracle as O

    mk = _maker()
    small = str(tmp_path / "small.h5")
    H.write_keras_h5(small, mk.params(31))
    r = subprocess.run([CONDA_PY, os.path.join(GOLDEN, "make_h5py_golden.py"), "verify", small, "31"], capture_output=True, text=True)
    assert r.returncode == 0 and '"datasets": 29' in r.stdout, r.stderr[-2000:]
    params = O.init_params(9, 27, seed=7, dtype=np.float32)
    full, npz = str(tmp_path / "result_w.h5"), str(tmp_path / "params.npz")
    H.write_keras_h5(full, params)
    np.savez(npz, **params)
    r = subprocess.run([CONDA_PY, os.path.join(GOLDEN, "make_h5py_golden.py"), "verify", full, npz], capture_output=True, text=True)
    assert r.returncode == 0 and '"datasets": %d' % len(params) in r.stdout, r.stderr[-2000:]
    # and h5py writing the fixture again gives what is committed (content, not bytes)
    again = str(tmp_path / "again.h5")
    assert subprocess.run([CONDA_PY, os.path.join(GOLDEN, "make_h5py_golden.py"), "write", again], capture_output=True).returncode == 0
    a, b = H.read_h5(again), H.read_h5(os.path.join(GOLDEN, "h5py_keras_layout.h5"))
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)


def test_reader_parses_many_group_file_written_by_h5py():
    """tests/golden/h5py_many_groups.h5 (make_h5py_golden.py write_many): 73 top-level layer groups and 64 nested groups under `model`, tiny
    datasets.  libhdf5 keeps at most 8 entries per symbol-table node, so both the root group and `model` are B-trees over SEVERAL nodes --
    the layout a full Keras weight file has (71 layers + 62 nested backbone layers) and the reduced fixture above never reaches."""
    mk = _maker()
    want = mk.many_group_items(40)
    path = os.path.join(GOLDEN, "h5py_many_groups.h5")
    got = {k.lstrip("/"): v for k, v in H.read_h5(path).items()}   # read_h5 keys are absolute paths
    assert len(want) == 200 and set(got) == set(want)
    assert all(got[k].dtype == np.float32 and got[k].shape == want[k].shape and np.array_equal(got[k], want[k]) for k in want)
    attrs = H.read_attrs(path)
    tops = sorted({k.split("/")[0] for k in want})
    assert len(tops) == mk.MANY_TOP and attrs["/"]["layer_names"] == [t.encode() for t in tops]
    assert len(attrs["/model"]["weight_names"]) == 2 * mk.MANY_NESTED
    assert attrs["/layer_071"]["weight_names"] == [b"layer_071/kernel:0"]
