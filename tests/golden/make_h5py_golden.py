#!/opt/conda/bin/python3.9
"""Golden vectors from the REAL HDF5 library for the Keras weight-file path (SURVEY 8(f) rank 1, round-2 verdict row f1).

The image's conda environment (/opt/conda, Python 3.9) ships h5py 3.3.0 on libhdf5 1.10.6 -- the library Keras' `save_weights` /
`load_weights` go through.  (TensorFlow / Keras themselves are absent, so what this pins is the FILE FORMAT as libhdf5 writes and reads it,
with Keras' group / dataset / attribute layout restated from keras/saving/hdf5_format.py: `save_weights_to_hdf5_group`,
`save_attributes_to_hdf5_group`; the variable ORDER inside the nested backbone stays an assumption, h5_weights.keras_backbone_layer_order.)

    /opt/conda/bin/python3.9 tests/golden/make_h5py_golden.py write  tests/golden/h5py_keras_layout.h5   # fixture: h5py writes, our reader parses
    /opt/conda/bin/python3.9 tests/golden/make_h5py_golden.py write_many tests/golden/h5py_many_groups.h5    # 73 top-level + 64 nested groups: multi-node group B-trees
    /opt/conda/bin/python3.9 tests/golden/make_h5py_golden.py verify <file written by h5_weights.write_keras_h5> <seed | params.npz>   # h5py reads OUR writer

`write` stores a reduced layer set (the full network is 59 MB): backbone layers nested under `model` (conv0, bn_data, bn0, one residual
unit), one CLADE layer with its inner sync_batch_normalization scope, one PartialConvolution, one plain Conv2D head -- every dataset
seeded by (seed, name) so the test regenerates the expected arrays without h5py.  Runs under the conda interpreter only (NumPy + h5py; this
file does not import the package, whose __init__ needs PyTorch)."""
import importlib.util
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# '<layer>.<field>' -> shape of the reduced parameter set
SHAPES = {
    "bn_data.beta": (3,), "bn_data.moving_mean": (3,), "bn_data.moving_variance": (3,),
    "conv0.kernel": (7, 7, 3, 64),
    "bn0.gamma": (64,), "bn0.beta": (64,), "bn0.moving_mean": (64,), "bn0.moving_variance": (64,),
    "stage1_unit1_bn1.gamma": (64,), "stage1_unit1_bn1.beta": (64,), "stage1_unit1_bn1.moving_mean": (64,), "stage1_unit1_bn1.moving_variance": (64,),
    "stage1_unit1_sc.kernel": (1, 1, 64, 64), "stage1_unit1_conv1.kernel": (3, 3, 64, 64),
    "stage1_unit1_bn2.gamma": (64,), "stage1_unit1_bn2.beta": (64,), "stage1_unit1_bn2.moving_mean": (64,), "stage1_unit1_bn2.moving_variance": (64,),
    "stage1_unit1_conv2.kernel": (3, 3, 64, 64),
    "pv_block_10_prepare_conv2d.weights": (35, 3, 3, 32),
    "pv_block_10_clade.gamma": (9, 32), "pv_block_10_clade.beta": (9, 32), "pv_block_10_clade.moving_mean": (32,), "pv_block_10_clade.moving_variance": (32,),
    "pv_block_5_bn.gamma": (32,), "pv_block_5_bn.beta": (32,), "pv_block_5_bn.moving_mean": (32,), "pv_block_5_bn.moving_variance": (32,),
    "pv_final_conv_vertex.kernel": (1, 1, 32, 27),
}


def tensor(seed, name, shape):
    """float32 values seeded by (seed, crc32(name)): reproducible in any NumPy >= 1.17"""
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())]).standard_normal(shape).astype(np.float32)


def params(seed):
    return {k: tensor(seed, k, s) for k, s in SHAPES.items()}


def h5w():
    """casapose_amd/utils/h5_weights.py loaded by path (NumPy + struct only)"""
    spec = importlib.util.spec_from_file_location("h5_weights", os.path.join(ROOT, "casapose_amd", "utils", "h5_weights.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def write(path, seed=20):
    """what keras.saving.hdf5_format.save_weights_to_hdf5_group does, with h5py: layer groups, `weight_names` attribute (NumPy 'S' array),
    one dataset per variable created with create_dataset(name, shape, dtype) and filled -- h5py's defaults (contiguous layout, earliest libver)"""
    import h5py

    p = params(seed)
    datasets, attrs = h5w().keras_layout(p)
    with h5py.File(path, "w") as f:
        root = attrs[""]
        f.attrs["layer_names"] = np.asarray(root["layer_names"])            # bytes -> fixed-length 'S' array, as Keras passes it
        f.attrs["backend"] = root["backend"]
        f.attrs["keras_version"] = root["keras_version"]
        for layer in [n.decode() for n in root["layer_names"]]:
            g = f.create_group(layer)
            names = attrs[layer]["weight_names"]
            g.attrs["weight_names"] = np.asarray(names)
            for n in names:
                val = datasets["%s/%s" % (layer, n.decode())]
                d = g.create_dataset(n.decode(), val.shape, dtype=val.dtype)   # nested names create the intermediate groups, as in Keras
                if val.shape:
                    d[:] = val
                else:
                    d[()] = val
    print(json.dumps({"path": path, "seed": seed, "h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version, "datasets": len(datasets)}))


# ---- many groups: more children per group than one symbol-table node holds (libhdf5 splits a group's entries over several SNODs above 8)
MANY_TOP, MANY_NESTED = 73, 64


def many_group_items(seed=40):
    """path -> array for a Keras-LAYOUT file of tiny datasets: 72 top-level layer groups (one dataset each, nested under the layer's own name
    as Keras does: '<layer>/<layer>/kernel:0') plus one group `model` holding 64 nested layer groups of two datasets each."""
    items = {}
    for i in range(MANY_TOP - 1):
        name = "layer_%03d" % i
        items["%s/%s/kernel:0" % (name, name)] = tensor(seed, name, (1 + i % 3, 2))
    for i in range(MANY_NESTED):
        name = "stage_%03d_bn" % i
        items["model/%s/gamma:0" % name] = tensor(seed, name + "g", (3,))
        items["model/%s/beta:0" % name] = tensor(seed, name + "b", (3,))
    return items


def write_many(path, seed=40):
    import h5py

    items = many_group_items(seed)
    layers = sorted({k.split("/")[0] for k in items})
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = np.asarray([n.encode() for n in layers])
        f.attrs["backend"] = "tensorflow"
        f.attrs["keras_version"] = "2.9.0"
        for layer in layers:
            g = f.create_group(layer)
            names = sorted(k[len(layer) + 1:] for k in items if k.split("/")[0] == layer)
            g.attrs["weight_names"] = np.asarray([n.encode() for n in names])
            for n in names:
                val = items["%s/%s" % (layer, n)]
                g.create_dataset(n, val.shape, dtype=val.dtype)[:] = val
    print("wrote", path, os.path.getsize(path), "bytes,", len(layers), "top-level groups,", len(items), "datasets")


def verify(path, seed):
    """h5py / libhdf5 reads a file written by OUR writer (h5_weights.write_keras_h5 on params(seed), or on the arrays of an .npz when `seed`
    is a path): every dataset bit-equal, the attributes as Keras' loader reads them (load_attributes_from_hdf5_group:
    f.attrs['layer_names'], g.attrs['weight_names'] -> lists of bytes)."""
    import h5py

    p = dict(np.load(seed)) if isinstance(seed, str) else params(seed)
    datasets, attrs = h5w().keras_layout(p)
    with h5py.File(path, "r") as f:
        layer_names = [n.decode() if isinstance(n, bytes) else n for n in f.attrs["layer_names"]]
        assert layer_names == [n.decode() for n in attrs[""]["layer_names"]], layer_names
        assert f.attrs["backend"] in (b"tensorflow", "tensorflow")
        seen = 0
        for layer in layer_names:
            g = f[layer]
            names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs["weight_names"]]
            assert names == [n.decode() for n in attrs[layer]["weight_names"]], (layer, names)
            for n in names:
                got = np.asarray(g[n])                                   # Keras: weight_values = [np.asarray(g[weight_name]) for ...]
                want = datasets["%s/%s" % (layer, n)]
                assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got, want), (layer, n)
                seen += 1
        assert seen == len(datasets)
    print(json.dumps({"verified": path, "datasets": seen, "h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version}))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "write":
        write(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    elif len(sys.argv) >= 3 and sys.argv[1] == "write_many":
        write_many(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "verify":
        verify(sys.argv[2], int(sys.argv[3]) if sys.argv[3].isdigit() else sys.argv[3])
    else:
        sys.exit(__doc__)
