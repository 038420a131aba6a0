"""The model object handed back by the constructors: the subset of the Keras `Model`
surface the reference's scripts use (train_casapose.py:374-406,537,592,903;
test_casapose.py:228,235-238,299), backed by casapose_amd.engine.CasaposeNet.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ... import engine
from ... import train_engine
from ...utils import h5_weights


def _he_uniform(rng, shape, fan_in):
    lim = math.sqrt(6.0 / fan_in)
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def initial_parameters(seg_dim: int, ver_dim: int, dims: Sequence[int], seed: Optional[int] = None,
                       partial: Sequence[bool] = engine.PARTIAL_DEFAULT, pvnet: bool = False, shared: Sequence[bool] = (False,) * 5,
                       reuse_first: bool = False, skips2: bool = True) -> Dict[str, np.ndarray]:
    """Keras-default initial state: he_uniform kernels (resnet.py:31; _normalization_layers.py:317),
    BN gamma 1 / beta 0 / moving mean 0 / moving variance 1, CLADE gamma 1 / beta 0
    (_normalization_layers.py:96-107).  Keys are `<keras layer name>.<weight>`."""
    rng = np.random.default_rng(seed)
    p: Dict[str, np.ndarray] = {}

    def bn(name, c, gamma=True, beta=True):
        if gamma:
            p[name + ".gamma"] = np.ones(c, np.float32)
        if beta:
            p[name + ".beta"] = np.zeros(c, np.float32)
        p[name + ".moving_mean"] = np.zeros(c, np.float32)
        p[name + ".moving_variance"] = np.ones(c, np.float32)

    p["conv0.kernel"] = _he_uniform(rng, (7, 7, 3, 64), 147)
    bn("bn_data", 3, gamma=False)
    bn("bn0", 64)
    cin = 64
    for s, f in enumerate(engine.STAGE_FILTERS):
        for u in range(2):
            base = "stage%d_unit%d_" % (s + 1, u + 1)
            if u == 0:
                p[base + "sc.kernel"] = _he_uniform(rng, (1, 1, cin, f), cin)
            p[base + "conv1.kernel"] = _he_uniform(rng, (3, 3, cin, f), 9 * cin)
            p[base + "conv2.kernel"] = _he_uniform(rng, (3, 3, f, f), 9 * f)
            bn(base + "bn1", cin)
            bn(base + "bn2", f)
            cin = f
    bn("bn1", 512)
    dec_in = (512, dims[0] + 128, dims[1] + 64, dims[2] + 64, dims[3] + 3)
    for i in range(5):
        ci, co = dec_in[i], dims[i]
        if shared[i]:  # one PartialConvolution for blocks i+1 and i+6 (pose_models.py:727-731)
            p["pv_block_%d_%d_conv2d.weights" % (i + 1, i + 6)] = _he_uniform(rng, (ci, 3, 3, co), 9 * ci)
        else:
            p["pv_block_%d_conv2d.kernel" % (i + 1)] = _he_uniform(rng, (3, 3, ci, co), 9 * ci)
        bn("pv_block_%d_bn" % (i + 1), co)
        if pvnet:
            continue
        ci2 = ci if (skips2 or i == 0) else dims[i - 1]
        if shared[i] or (i == 0 and reuse_first):
            pass  # no convolution weights of its own
        elif partial[i]:
            p["pv_block_%d_prepare_conv2d.weights" % (i + 6)] = _he_uniform(rng, (ci2, 3, 3, co), 9 * ci2)
        else:  # ordinary pad + Conv2D in decoder 2 (casapose.py:69-74)
            p["pv_block_%d_conv2d.kernel" % (i + 6)] = _he_uniform(rng, (3, 3, ci2, co), 9 * ci2)
        bn("pv_block_%d_clade" % (i + 6), co, gamma=False, beta=False)
        p["pv_block_%d_clade.gamma" % (i + 6)] = np.ones((seg_dim, co), np.float32)
        p["pv_block_%d_clade.beta" % (i + 6)] = np.zeros((seg_dim, co), np.float32)
    if pvnet:  # PVNet: one head for segmentation + vector field (pose_models.py:678)
        p["pv_final_conv.kernel"] = _he_uniform(rng, (1, 1, dims[4], seg_dim + ver_dim), dims[4])
        return p
    p["pv_final_conv_segmentation.kernel"] = _he_uniform(rng, (1, 1, dims[4], seg_dim), dims[4])
    p["pv_final_conv_vertex.kernel"] = _he_uniform(rng, (1, 1, dims[4], ver_dim), dims[4])
    return p


class Layer:
    """Named view of one layer's weights (`net.get_layer(name).get_weights()/set_weights()`,
    train_casapose.py:403-406)."""

    def __init__(self, model: "CasaposeModel", name: str, keys: List[str]):
        self._model, self.name, self._keys = model, name, keys
        self.trainable = True

    def get_weights(self) -> List[np.ndarray]:
        self._model._sync_from_store()  # after a train_step the flat device store holds the current weights
        return [self._model._params[k].copy() for k in self._keys]

    def set_weights(self, weights: Sequence[np.ndarray]):
        self._model._sync_from_store()
        if len(weights) != len(self._keys):
            raise ValueError("layer %s expects %d arrays, got %d" % (self.name, len(self._keys), len(weights)))
        new = dict(self._model._params)
        for k, w in zip(self._keys, weights):
            w = np.asarray(w, dtype=np.float32)
            if w.shape != new[k].shape:
                raise ValueError("layer %s: weight %s has shape %s, expected %s" % (self.name, k, w.shape, new[k].shape))
            new[k] = w
        self._model.set_parameters(new)


class CasaposeModel:
    def __init__(self, name: str, ver_dim: int, seg_dim: int, dims: Sequence[int], input_shape=None,
                 input_segmentation_shape=None, weights=None, output_lablemap: bool = False, device=None, seed=None,
                 fuse_upsample: bool = True, fuse_heads: bool = True, partial: Sequence[bool] = engine.PARTIAL_DEFAULT,
                 guided: Sequence[bool] = engine.GUIDED_DEFAULT, bilinear: Sequence[bool] = engine.BILINEAR_DEFAULT, pvnet: bool = False,
                 shared: Sequence[bool] = (False,) * 5, reuse_first: bool = False, skips2: bool = True, conv_mode: Optional[str] = None,
                 f16x2_guard: Optional[bool] = None):
        self.output_lablemap = bool(output_lablemap)
        self.name = name
        self.ver_dim, self.seg_dim = int(ver_dim), int(seg_dim)
        self.input_shape = tuple(input_shape) if input_shape is not None else None
        self.input_segmentation_shape = tuple(input_segmentation_shape) if input_segmentation_shape is not None else None
        self.input_names = ["data"] + (["data_segmentation"] if self.input_segmentation_shape is not None else [])
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self._dims = tuple(dims)
        if isinstance(weights, str) and weights == "imagenet":
            warnings.warn("weights='imagenet': the reference downloads ImageNet ResNet-18 weights (weights.py:13-39); "
                          "no network here -- using he_uniform initialisation; call load_weights() for real weights")
            weights = None
        self._partial, self._guided = tuple(bool(v) for v in partial), tuple(bool(v) for v in guided)
        self._bilinear = tuple(bool(v) for v in bilinear)
        self._pvnet = bool(pvnet)
        if self._pvnet and self.input_segmentation_shape is not None:
            raise ValueError("PVNet has no data_segmentation input")
        self._sharing = dict(shared=tuple(bool(v) for v in shared), reuse_first=bool(reuse_first), skips2=bool(skips2))
        self._params = initial_parameters(self.seg_dim, self.ver_dim, self._dims, seed, self._partial, self._pvnet, **self._sharing)
        self._net = engine.CasaposeNet(self._params, self.seg_dim, self.ver_dim, self.device, self._dims, fuse_upsample, fuse_heads,
                                       self._partial, self._guided, bilinear=self._bilinear, pvnet=self._pvnet, conv_mode=conv_mode, f16x2_guard=f16x2_guard, **self._sharing)
        self._store: Optional[train_engine.ParamStore] = None   # training state (flat master weights + Adam moments)
        self._plan: Optional[train_engine.TrainPlan] = None
        self._params_stale = False                               # the store holds newer weights than self._params
        if isinstance(weights, str):
            self.load_weights(weights)
        self._layers = self._build_layers()

    # ---- training state --------------------------------------------------------------------------
    def training_plan(self, batch: int, h: int, w: int, group=None, world_size: int = 1):
        """The launch plan of the training step for this input shape (created on first use; the flat parameter
        store and its Adam moments survive shape changes)."""
        if self._store is None:
            self._store = train_engine.ParamStore(self._params, self.device)
        p = self._plan
        if p is None or (p.batch, p.h, p.w) != (batch, h, w) or p.group is not group:
            self._plan = train_engine.TrainPlan(self._store, self.seg_dim, self.ver_dim, batch, h, w, self._dims, group, world_size,
                                                self._partial, self._guided, bilinear=self._bilinear, pvnet=self._pvnet, **self._sharing)
            self._plan.refresh_weights(torch.cuda.current_stream(self.device).cuda_stream)
        return self._plan, self.device

    def mark_trained(self):
        self._params_stale = True

    def _sync_from_store(self):
        if self._params_stale and self._store is not None:
            self._params = {k: v.astype(np.float32) for k, v in self._store.export().items()}
            self._net.set_params(self._params)
            self._params_stale = False

    # ---- Keras-like surface ----------------------------------------------------------------
    def _build_layers(self) -> List[Layer]:
        groups: Dict[str, List[str]] = {}
        for k in self._params:
            groups.setdefault(k.split(".")[0], []).append(k)
        # Keras `layer.get_weights()` order: BatchNormalization gamma, beta, moving statistics; the CLADE layers add beta before
        # gamma (_normalization_layers.py:96-107), so positional set_weights() from reference-ordered lists lines up
        order = h5_weights.KERAS_FIELD_ORDER
        clade = dict(order, beta=1, gamma=2)
        return [Layer(self, n, sorted(ks, key=lambda k: (clade if n.endswith("_clade") else order)[k.split(".")[1]])) for n, ks in groups.items()]

    @property
    def layers(self) -> List[Layer]:
        return self._layers

    def get_layer(self, name: str) -> Layer:
        for l in self._layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: %s" % name)

    @property
    def trainable_variables(self) -> List[str]:
        frozen = {l.name for l in self._layers if not l.trainable}
        return [k for k in self._params if not k.endswith(("moving_mean", "moving_variance")) and k.split(".")[0] not in frozen]

    def count_params(self) -> int:
        return int(sum(v.size for v in self._params.values()))

    def summary(self, print_fn=print):
        print_fn('Model: "%s"  (MI355X / gfx950 backend)' % self.name)
        for l in self._layers:
            shapes = ", ".join("%s%s" % (k.split(".")[1], tuple(self._params[k].shape)) for k in l._keys)
            print_fn("  %-36s %s" % (l.name, shapes))
        print_fn("Total params: {:,}".format(self.count_params()))

    def get_parameters(self) -> Dict[str, np.ndarray]:
        self._sync_from_store()
        return {k: v.copy() for k, v in self._params.items()}

    def set_parameters(self, params: Dict[str, np.ndarray]):
        missing = set(self._params) - set(params)
        if missing:
            raise ValueError("missing parameters: %s" % sorted(missing)[:5])
        for k, v in self._params.items():
            if tuple(np.shape(params[k])) != v.shape:
                raise ValueError("parameter %s has shape %s, expected %s" % (k, np.shape(params[k]), v.shape))
        self._params = {k: np.asarray(params[k], dtype=np.float32) for k in self._params}
        self._net.set_params(self._params)
        if getattr(self, "_store", None) is not None:  # keep the training copy (but not its Adam moments) in step
            for k in self._store.offsets:
                self._store.view(k).copy_(torch.from_numpy(self._params[k]))
            for k, t in self._store.state.items():
                t.copy_(torch.from_numpy(self._params[k]))
            if self._plan is not None:
                self._plan.refresh_weights(torch.cuda.current_stream(self.device).cuda_stream)
            self._params_stale = False

    def save_weights(self, path: str):
        """`net.save_weights(frozen_path + "/result_w.h5")` (train_casapose.py:903): a path ending in .h5 / .hdf5 / .keras gets a
        real HDF5 file with Keras' group tree, dataset names and `layer_names` / `weight_names` attributes
        (utils/h5_weights.write_keras_h5), laid out for Keras' load_weights(by_name=True) (variable order of the nested backbone as derived in h5_weights.keras_backbone_layer_order; not verifiable without Keras here) and read by load_weights() below; any other
        extension stores the '<layer>.<field>' -> array mapping as .npz."""
        self._sync_from_store()
        if str(path).lower().endswith((".h5", ".hdf5", ".keras")):
            h5_weights.write_keras_h5(path, self._params)
            return
        with open(path, "wb") as f:
            np.savez(f, **self._params)

    def load_weights(self, path: str, by_name: bool = True, skip_mismatch: bool = True):
        """by_name / skip_mismatch follow test_casapose.py:225-228: unknown names are ignored and
        shape mismatches are skipped (with a warning) instead of raising."""
        self._sync_from_store()
        if h5_weights.is_hdf5(path):  # a Keras save_weights file (the reference's result_w_8.h5 / result_w_13.h5)
            found = h5_weights.keras_weights_from_h5(path, {k.split(".")[0] for k in self._params})

            class _Npz:  # same access pattern as np.load
                files = list(found)

                def __getitem__(self, k):
                    return found[k]

            data = _Npz()
        else:
            data = np.load(path)
        new = dict(self._params)
        for k in data.files:
            if k not in new:
                continue
            if data[k].shape != new[k].shape:
                if skip_mismatch:
                    warnings.warn("load_weights: skipping %s, shape %s != %s" % (k, data[k].shape, new[k].shape))
                    continue
                raise ValueError("load_weights: %s has shape %s, expected %s" % (k, data[k].shape, new[k].shape))
            new[k] = data[k].astype(np.float32)
        self.set_parameters(new)

    # ---- forward -----------------------------------------------------------------------------
    def _to_device(self, x) -> torch.Tensor:
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def __call__(self, inputs, training: bool = False) -> torch.Tensor:
        """inputs: [img] or [img, seg_onehot] (NHWC float32, torch or numpy).  Returns the
        device tensor [B,H,W,seg_dim+ver_dim] = concat(seg logits, vertex) (pose_models.py:628)."""
        if isinstance(inputs, (torch.Tensor, np.ndarray)):
            inputs = [inputs]
        if len(inputs) != len(self.input_names):
            raise ValueError("model %s expects inputs %s, got %d tensors" % (self.name, self.input_names, len(inputs)))
        img = self._to_device(inputs[0])
        seg = self._to_device(inputs[1]) if len(inputs) > 1 else None
        if self.input_shape is not None and tuple(img.shape[1:]) != self.input_shape:
            raise ValueError("input `data` has shape %s, model was built for %s" % (tuple(img.shape[1:]), self.input_shape))
        if training:
            # batch-statistics forward on the training plan (net(net_input, training=True), train_casapose.py:537);
            # the gradient side is driven by casapose_amd.training.train_step
            plan, _ = self.training_plan(img.shape[0], img.shape[1], img.shape[2])
            cond = torch.argmax(seg, dim=-1).to(torch.uint8).contiguous() if seg is not None else None
            return plan.forward(img, cond)
        self._sync_from_store()
        out = self._net.forward(img, seg)
        if self.output_lablemap:
            # soft-argmax head (pose_models.py:619-626): sum_k softmax(1e6 * logits)_k * k -- with the saturated softmax this is the
            # arg-max index as a float; the model then returns [label | vertex] instead of [logits | vertex]
            k = self.seg_dim
            lab = torch.argmax(out[..., :k], dim=3, keepdim=True).to(out.dtype)
            return torch.cat([lab, out[..., k:]], dim=3)
        return out

    def predict(self, inputs):
        return self(inputs, training=False)
