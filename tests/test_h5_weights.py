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
