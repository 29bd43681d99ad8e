"""Policy/value residual CNN of reversi_zero/agent/model.py:28-72, as (a) a PyTorch module with the
Keras graph's exact layer semantics — the fp32 reference the 1e-5 tolerance is measured against and
the optional `torch` evaluation backend — and (b) the folded weight blob ("raznet v1") that both
the HIP forward kernels (csrc/raz_net.hip) and the CPU oracle (oracle/orc_net.c) consume.

Keras defaults restated: Conv2D use_bias=True; 3x3 convs padding="same", 1x1 head
convs "valid"; BatchNormalization(axis=1) eps=1e-3, inference uses moving statistics;
Flatten on channels_first => index c*64 + y*8 + x; Dense kernels stored (in, out); policy head
softmax, value head tanh.  Initialisation = Keras defaults (glorot_uniform kernels, zero biases,
BN gamma=1 beta=0 mean=0 var=1) from a seeded torch.Generator; benches use random-init weights of the
right architecture (no trained weight file travels with the repository).  Keras weight files are read
and written through lib/keras_h5.py (no h5py needed).
"""
import hashlib
import json
import os
import struct

import numpy as np
import torch
from torch import nn

RAZNET_MAGIC = 0x4E5A4152  # "RAZN"
BN_EPS = 1e-3              # keras BatchNormalization default epsilon


def _glorot_uniform_(w, fan_in, fan_out, gen):
    limit = (6.0 / (fan_in + fan_out)) ** 0.5
    with torch.no_grad():
        w.copy_((torch.rand(w.shape, generator=gen, dtype=torch.float32) * 2 - 1) * limit)


class _ConvBN(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=BN_EPS)

    def forward(self, x):
        return self.bn(self.conv(x))


class ReversiNet(nn.Module):
    """torch restatement of the Keras graph (agent/model.py:28-72).  Always used in eval() mode."""

    def __init__(self, filters=256, res_layers=10, value_fc=256, filter_size=3):
        super().__init__()
        self.filters, self.res_layers, self.value_fc, self.filter_size = filters, res_layers, value_fc, filter_size
        self.stem = _ConvBN(2, filters, filter_size)
        self.res = nn.ModuleList(nn.ModuleList([_ConvBN(filters, filters, filter_size),
                                                _ConvBN(filters, filters, filter_size)])
                                 for _ in range(res_layers))
        self.policy_conv = _ConvBN(filters, 2, 1)
        self.policy_fc = nn.Linear(128, 64)
        self.value_conv = _ConvBN(filters, 1, 1)
        self.value_fc1 = nn.Linear(64, value_fc)
        self.value_fc2 = nn.Linear(value_fc, 1)
        self.eval()

    def keras_init_(self, seed=0):
        gen = torch.Generator().manual_seed(seed)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                k = m.kernel_size[0] * m.kernel_size[1]
                _glorot_uniform_(m.weight, k * m.in_channels, k * m.out_channels, gen)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                _glorot_uniform_(m.weight, m.in_features, m.out_features, gen)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
        return self

    def randomize_bn_(self, seed=1):
        """Give BN layers non-trivial statistics (tests: exercises the folding)."""
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=gen))
                    m.bias.copy_(torch.rand(m.bias.shape, generator=gen) - 0.5)
                    m.running_mean.copy_(0.2 * (torch.rand(m.bias.shape, generator=gen) - 0.5))
                    m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=gen))
        return self

    def forward(self, x):
        """x: (N,2,8,8) float32 planes [own, enemy] -> (policy (N,64) softmax, value (N,1) tanh)."""
        x = torch.relu(self.stem(x))
        for c1, c2 in self.res:
            x = torch.relu(c2(torch.relu(c1(x))) + x)
        p = torch.relu(self.policy_conv(x)).flatten(1)
        p = torch.softmax(self.policy_fc(p), dim=1)
        v = torch.relu(self.value_conv(x)).flatten(1)
        v = torch.tanh(self.value_fc2(torch.relu(self.value_fc1(v))))
        return p, v

    # ---- folded blob -------------------------------------------------------------------------
    @staticmethod
    def _fold(cb):
        """conv+BN -> (w', b') in float64, rounded once to float32."""
        w = cb.conv.weight.detach().double()
        b = cb.conv.bias.detach().double()
        s = cb.bn.weight.detach().double() / torch.sqrt(cb.bn.running_var.detach().double() + cb.bn.eps)
        w = w * s.view(-1, 1, 1, 1)
        b = (b - cb.bn.running_mean.detach().double()) * s + cb.bn.bias.detach().double()
        return w.float().contiguous(), b.float().contiguous()

    def to_blob(self) -> bytes:
        """raznet v1: int32[8] header {magic, version=1, F, R, V, filter_size, 0, 0} then float32:
        conv0 w[F][2][3][3] b[F]; per block c1 w[F][F][3][3] b[F], c2 w b; policy conv w[2][F] b[2];
        policy dense W[128][64] (in,out) b[64]; value conv w[1][F] b[1]; dense1 W[64][V] b[V];
        dense2 W[V] b[1].  BN already folded."""
        if self.filter_size != 3:
            raise ValueError("raznet v1 supports cnn_filter_size=3 only (all shipped configs)")
        parts = []
        for cb in [self.stem] + [c for blk in self.res for c in blk]:
            w, b = self._fold(cb)
            parts += [w.reshape(-1), b]
        w, b = self._fold(self.policy_conv)
        parts += [w.reshape(-1), b, self.policy_fc.weight.detach().t().contiguous().reshape(-1),
                  self.policy_fc.bias.detach()]
        w, b = self._fold(self.value_conv)
        parts += [w.reshape(-1), b, self.value_fc1.weight.detach().t().contiguous().reshape(-1),
                  self.value_fc1.bias.detach(), self.value_fc2.weight.detach().reshape(-1),
                  self.value_fc2.bias.detach()]
        flat = torch.cat([p.float().reshape(-1) for p in parts]).cpu().numpy().astype(np.float32)
        assert flat.size == blob_float_count(self.filters, self.res_layers, self.value_fc)
        hdr = struct.pack("<8i", RAZNET_MAGIC, 1, self.filters, self.res_layers, self.value_fc, 3, 0, 0)
        return hdr + flat.tobytes()


def blob_float_count(F, R, V):
    return (F * 18 + F) + R * 2 * (F * F * 9 + F) + (2 * F + 2) + (128 * 64 + 64) + (F + 1) + (64 * V + V) + (V + 1)


def macs_per_position(F, R, V):
    """Multiply-accumulates of one forward pass (325,648 mini; 755,343,616 for 256x10)."""
    return 64 * (F * 18 + R * 2 * F * F * 9 + 2 * F + F) + 128 * 64 + 64 * V + V


# ---- weight interchange with the reference's Keras files --------------------------------------------------------
# The reference stores model_*_weight.h5 (Keras 2.1 save_weights, HDF5) + a model config JSON (agent/model.py:82-101).
# lib/keras_h5.py reads and writes those files without h5py.  In memory the interchange form is a dict of Keras-NAMED
# arrays: one array per Keras weight, key "<layer name>/<weight name>" exactly as `w.name` gives it on the reference
# side (e.g. "conv2d_3/kernel:0", "batch_normalization_3/moving_variance:0", "policy_out/bias:0"), arrays in Keras
# layouts (Conv2D kernel (kh, kw, in, out), Dense kernel (in, out)).  The same dict saved with np.savez is the ".npz
# bridge" of earlier versions (still read; tools/keras_npz_bridge.py converts on a machine with keras).  Layers are
# identified by kind, creation number (the numeric suffix Keras appends; any offset) and shape, never by absolute
# names or by their order in the file.
def _suffix(name):
    import re
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def keras_named_arrays(net):
    """ReversiNet -> {keras weight name: array} in Keras layouts, layers named as a fresh Keras session names them
    for agent/model.py:28-58 (conv2d_1.., batch_normalization_1.., dense_1, policy_out, value_out)."""
    out = {}
    convbns = [net.stem] + [c for blk in net.res for c in blk] + [net.policy_conv, net.value_conv]
    for i, cb in enumerate(convbns, start=1):
        out[f"conv2d_{i}/kernel:0"] = cb.conv.weight.detach().permute(2, 3, 1, 0).contiguous().numpy()
        out[f"conv2d_{i}/bias:0"] = cb.conv.bias.detach().numpy()
        out[f"batch_normalization_{i}/gamma:0"] = cb.bn.weight.detach().numpy()
        out[f"batch_normalization_{i}/beta:0"] = cb.bn.bias.detach().numpy()
        out[f"batch_normalization_{i}/moving_mean:0"] = cb.bn.running_mean.detach().numpy()
        out[f"batch_normalization_{i}/moving_variance:0"] = cb.bn.running_var.detach().numpy()
    for name, lin in (("policy_out", net.policy_fc), ("dense_1", net.value_fc1), ("value_out", net.value_fc2)):
        out[f"{name}/kernel:0"] = lin.weight.detach().t().contiguous().numpy()
        out[f"{name}/bias:0"] = lin.bias.detach().numpy()
    return out


def net_from_keras_named_arrays(arrays):
    """{keras weight name: array} -> ReversiNet.  Raises ValueError naming what is missing or mis-shaped."""
    layers = {}
    for key, a in arrays.items():
        lname, _, wname = key.partition("/")
        layers.setdefault(lname, {})[wname.split(":")[0].split("/")[-1]] = np.asarray(a)
    convs = sorted((n for n in layers if "kernel" in layers[n] and layers[n]["kernel"].ndim == 4), key=_suffix)
    bns = sorted((n for n in layers if "moving_variance" in layers[n]), key=_suffix)
    denses = [n for n in layers if "kernel" in layers[n] and layers[n]["kernel"].ndim == 2]
    if len(convs) < 3 or len(convs) != len(bns) or (len(convs) - 3) % 2 or len(denses) != 3:
        raise ValueError(f"not a reversi_zero model: {len(convs)} conv, {len(bns)} batch-norm, {len(denses)} dense layers")
    F, ks = layers[convs[0]]["kernel"].shape[3], layers[convs[0]]["kernel"].shape[0]
    R = (len(convs) - 3) // 2
    heads = convs[-2:]
    pol_conv = next((n for n in heads if layers[n]["kernel"].shape[3] == 2), None)
    val_conv = next((n for n in heads if layers[n]["kernel"].shape[3] == 1), None)
    if pol_conv is None or val_conv is None:
        raise ValueError("policy (2 filters) / value (1 filter) head convolutions not found")
    bn_of = dict(zip(convs, bns))   # Keras creates each BatchNormalization right after its Conv2D: same order
    pol_fc = next((n for n in denses if layers[n]["kernel"].shape == (128, 64)), None)
    v2 = next((n for n in denses if layers[n]["kernel"].shape[1] == 1), None)
    v1 = next((n for n in denses if n not in (pol_fc, v2)), None)
    if pol_fc is None or v1 is None or v2 is None or layers[v1]["kernel"].shape[0] != 64:
        raise ValueError("dense layers 128->64 (policy), 64->V, V->1 (value) not found")
    V = layers[v1]["kernel"].shape[1]
    net = ReversiNet(F, R, V, ks)
    convbns = [net.stem] + [c for blk in net.res for c in blk]
    with torch.no_grad():
        for cb, cname in list(zip(convbns, convs[:-2])) + [(net.policy_conv, pol_conv), (net.value_conv, val_conv)]:
            L, B = layers[cname], layers[bn_of[cname]]
            k = torch.from_numpy(np.ascontiguousarray(L["kernel"], dtype=np.float32)).permute(3, 2, 0, 1)
            if tuple(k.shape) != tuple(cb.conv.weight.shape):
                raise ValueError(f"{cname}: kernel shape {tuple(L['kernel'].shape)} does not fit {tuple(cb.conv.weight.shape)}")
            cb.conv.weight.copy_(k)
            cb.conv.bias.copy_(torch.from_numpy(np.asarray(L["bias"], dtype=np.float32)))
            cb.bn.weight.copy_(torch.from_numpy(np.asarray(B["gamma"], dtype=np.float32)))
            cb.bn.bias.copy_(torch.from_numpy(np.asarray(B["beta"], dtype=np.float32)))
            cb.bn.running_mean.copy_(torch.from_numpy(np.asarray(B["moving_mean"], dtype=np.float32)))
            cb.bn.running_var.copy_(torch.from_numpy(np.asarray(B["moving_variance"], dtype=np.float32)))
        for lin, name in ((net.policy_fc, pol_fc), (net.value_fc1, v1), (net.value_fc2, v2)):
            lin.weight.copy_(torch.from_numpy(np.ascontiguousarray(layers[name]["kernel"], dtype=np.float32)).t())
            lin.bias.copy_(torch.from_numpy(np.asarray(layers[name]["bias"], dtype=np.float32)))
    return net.eval()


def keras_layers(net):
    """The layers of the reference's graph (agent/model.py:28-72) for `net`'s architecture, in the order of Keras'
    `model.layers`, as [(layer name, class name, inbound layer names)], named as a fresh Keras session names them.

    Keras 2.1.2 (keras/engine/topology.py Container.__init__, restated - Keras is not part of the reference tree) sorts
    layers by depth = longest path to an output, deepest first, ties in the order a depth-first walk from the outputs
    (policy_out, then value_out) first meets them.  The trunk is a chain of distinct depths (creation order); the two
    heads tie: the value head is one Dense longer, so its layers sit one level deeper than the policy head's."""
    R = net.res_layers
    out = [("input_1", "InputLayer", []), ("conv2d_1", "Conv2D", ["input_1"]),
           ("batch_normalization_1", "BatchNormalization", ["conv2d_1"]), ("activation_1", "Activation", ["batch_normalization_1"])]
    for k in range(1, R + 1):
        a, b, prev = 2 * k, 2 * k + 1, f"activation_{2 * k - 1}"
        out += [(f"conv2d_{a}", "Conv2D", [prev]), (f"batch_normalization_{a}", "BatchNormalization", [f"conv2d_{a}"]),
                (f"activation_{a}", "Activation", [f"batch_normalization_{a}"]), (f"conv2d_{b}", "Conv2D", [f"activation_{a}"]),
                (f"batch_normalization_{b}", "BatchNormalization", [f"conv2d_{b}"]),
                (f"add_{k}", "Add", [prev, f"batch_normalization_{b}"]), (f"activation_{b}", "Activation", [f"add_{k}"])]
    p, v, trunk = 2 * R + 2, 2 * R + 3, f"activation_{2 * R + 1}"
    out += [(f"conv2d_{v}", "Conv2D", [trunk]),                                                                  # depth 5
            (f"conv2d_{p}", "Conv2D", [trunk]), (f"batch_normalization_{v}", "BatchNormalization", [f"conv2d_{v}"]),   # 4
            (f"batch_normalization_{p}", "BatchNormalization", [f"conv2d_{p}"]),
            (f"activation_{v}", "Activation", [f"batch_normalization_{v}"]),                                     # 3
            (f"activation_{p}", "Activation", [f"batch_normalization_{p}"]), ("flatten_2", "Flatten", [f"activation_{v}"]),   # 2
            ("flatten_1", "Flatten", [f"activation_{p}"]), ("dense_1", "Dense", ["flatten_2"]),                  # 1
            ("policy_out", "Dense", ["flatten_1"]), ("value_out", "Dense", ["dense_1"])]                         # 0
    return out


_WEIGHTS_OF = {"Conv2D": ("kernel:0", "bias:0"), "Dense": ("kernel:0", "bias:0"),
               "BatchNormalization": ("gamma:0", "beta:0", "moving_mean:0", "moving_variance:0")}


def keras_weight_layers(net):
    """[(layer name, [(weight name, array), ...])] for lib/keras_h5.write_keras_weights: every layer of keras_layers(net),
    the weightless ones with an empty list (Keras' save_weights lists them too)."""
    arrays = keras_named_arrays(net)
    return [(name, [(f"{name}/{w}", arrays[f"{name}/{w}"]) for w in _WEIGHTS_OF.get(cls, ())]) for name, cls, _ in keras_layers(net)]


def keras_model_config(net, l2_reg=1e-4):
    """What Keras 2.1.2 `Model.get_config()` returns for the reference's graph (agent/model.py:28-58) - the JSON the
    reference writes next to the weights and feeds to `Model.from_config` (agent/model.py:86,96).  Restated from Keras'
    layer `get_config` methods; Keras is absent here, so this is checked for structure only (tests/test_keras_h5.py)."""
    l2 = float(np.float32(l2_reg))   # keras.regularizers.L1L2 stores K.cast_to_floatx(l2)
    glorot = {"class_name": "VarianceScaling", "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}}
    zeros, ones = {"class_name": "Zeros", "config": {}}, {"class_name": "Ones", "config": {}}
    reg = {"class_name": "L1L2", "config": {"l1": 0.0, "l2": l2}}

    def conv(name, filters, k, padding):
        return {"name": name, "trainable": True, "filters": filters, "kernel_size": [k, k], "strides": [1, 1], "padding": padding,
                "data_format": "channels_first", "dilation_rate": [1, 1], "activation": "linear", "use_bias": True,
                "kernel_initializer": glorot, "bias_initializer": zeros, "kernel_regularizer": reg, "bias_regularizer": None,
                "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}

    def dense(name, units, activation):
        return {"name": name, "trainable": True, "units": units, "activation": activation, "use_bias": True,
                "kernel_initializer": glorot, "bias_initializer": zeros, "kernel_regularizer": reg, "bias_regularizer": None,
                "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}

    def bn(name):
        return {"name": name, "trainable": True, "axis": 1, "momentum": 0.99, "epsilon": BN_EPS, "center": True, "scale": True,
                "beta_initializer": zeros, "gamma_initializer": ones, "moving_mean_initializer": zeros,
                "moving_variance_initializer": ones, "beta_regularizer": None, "gamma_regularizer": None,
                "beta_constraint": None, "gamma_constraint": None}

    F, R, V, ks = net.filters, net.res_layers, net.value_fc, net.filter_size
    heads = {f"conv2d_{2 * R + 2}": (2, 1, "valid"), f"conv2d_{2 * R + 3}": (1, 1, "valid")}
    layers = []
    for name, cls, inbound in keras_layers(net):
        if cls == "InputLayer":
            cfg = {"batch_input_shape": [None, 2, 8, 8], "dtype": "float32", "sparse": False, "name": name}
        elif cls == "Conv2D":
            cfg = conv(name, *heads.get(name, (F, ks, "same")))
        elif cls == "BatchNormalization":
            cfg = bn(name)
        elif cls == "Activation":
            cfg = {"name": name, "trainable": True, "activation": "relu"}
        elif cls == "Dense":
            cfg = dense(name, *{"policy_out": (64, "softmax"), "dense_1": (V, "relu"), "value_out": (1, "tanh")}[name])
        else:   # Add, Flatten
            cfg = {"name": name, "trainable": True}
        layers.append({"name": name, "class_name": cls, "config": cfg,
                       "inbound_nodes": [[[i, 0, 0, {}] for i in inbound]] if inbound else []})
    return {"name": "reversi_model", "layers": layers, "input_layers": [["input_1", 0, 0]],
            "output_layers": [["policy_out", 0, 0], ["value_out", 0, 0]]}


HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"
NPZ_MAGIC = b"PK"


class ReversiModel:
    """Same role and method names as the reference's ReversiModel (agent/model.py:22-101).  `self.model` is the torch
    module.  Weight files, told apart by their first bytes whatever the (reference-fixed) file name says:
      * a Keras HDF5 weight file (`save_weights`, agent/model.py:94-101) - what the reference's workers write and read;
        read and written through lib/keras_h5.py;
      * a Keras-named .npz (see above): the bridge format of earlier versions, still read;
      * a torch state-dict (what save() wrote before either existed)."""

    def __init__(self, config):
        self.config = config
        self.model = None  # type: ReversiNet
        self.digest = None

    def build(self, seed=0):
        mc = self.config.model
        self.model = ReversiNet(mc.cnn_filter_num, mc.res_layer_num, mc.value_fc_size,
                                mc.cnn_filter_size).keras_init_(seed)

    @staticmethod
    def fetch_digest(weight_path):
        if os.path.exists(weight_path):
            m = hashlib.sha256()
            with open(weight_path, "rb") as f:
                m.update(f.read())
            return m.hexdigest()

    def load(self, config_path, weight_path):
        """agent/model.py:82-92.  The architecture is taken from the weights themselves (layer kinds and shapes), so the
        Keras config JSON is only required to exist, as in the reference."""
        if not (os.path.exists(config_path) and os.path.exists(weight_path)):
            return False
        with open(weight_path, "rb") as f:
            magic = f.read(8)
        if magic == HDF5_MAGIC:
            from ..lib.keras_h5 import read_keras_weights
            arrays, _ = read_keras_weights(weight_path)
            self.model = net_from_keras_named_arrays(arrays)
        elif magic[:2] == NPZ_MAGIC and not _is_torch_zip(weight_path):
            with np.load(weight_path) as z:
                self.model = net_from_keras_named_arrays({k: z[k] for k in z.files})
        else:   # a torch state-dict: the architecture is in the tensor shapes
            try:
                sd = torch.load(weight_path, map_location="cpu")
                w0 = sd["stem.conv.weight"]
                blocks = {int(k.split(".")[1]) for k in sd if k.startswith("res.")}
                self.model = ReversiNet(w0.shape[0], len(blocks), sd["value_fc1.weight"].shape[0], w0.shape[2])
                self.model.load_state_dict(sd)
            except Exception as e:
                raise ValueError(f"{weight_path} is neither a Keras HDF5 weight file, a Keras-named .npz nor a torch "
                                 f"state-dict of ReversiNet ({type(e).__name__}: {e})")
            self.model.eval()
        self.digest = self.fetch_digest(weight_path)
        return True

    def save(self, config_path, weight_path, weight_format="h5"):
        """agent/model.py:94-101: the Keras model config JSON (`get_config()`) + the Keras weight file (`save_weights`),
        both in the reference's formats, so that its opt / eval workers and GUI load what this package saved.
        weight_format="npz" writes the Keras-named .npz bridge instead."""
        m = self.model
        l2 = getattr(getattr(self.config, "model", None), "l2_reg", 1e-4)
        with open(config_path, "wt") as f:
            json.dump(keras_model_config(m, l2), f)
        tmp = weight_path + ".tmp"
        if weight_format == "h5":
            from ..lib.keras_h5 import write_keras_weights
            write_keras_weights(tmp, keras_weight_layers(m))
        elif weight_format == "npz":
            np.savez(tmp + ".npz", **keras_named_arrays(m))
            tmp += ".npz"
        else:
            raise ValueError(f"weight_format {weight_format!r}: 'h5' or 'npz'")
        os.replace(tmp, weight_path)   # readers poll the digest (api.py:117-125): never expose a half-written file
        self.digest = self.fetch_digest(weight_path)


def _is_torch_zip(path):
    """torch.save also writes a zip archive: it holds a `*/data.pkl` member, an .npz holds only `*.npy` members."""
    import zipfile
    try:
        with zipfile.ZipFile(path) as z:
            return any(n.endswith("data.pkl") for n in z.namelist())
    except zipfile.BadZipFile:
        return False
