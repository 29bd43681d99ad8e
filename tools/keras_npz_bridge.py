#!/usr/bin/env python
"""tools/keras_npz_bridge.py - weight interchange between the reference's Keras files and this package THROUGH KERAS.
The package reads and writes Keras h5 files by itself (lib/keras_h5.py); this script is the fallback that lets Keras do
the conversion.  RUNS ON THE REFERENCE SIDE (needs keras + h5py, which this repo's environment does not have):

    python tools/keras_npz_bridge.py h5-to-npz model_config.json model_weight.h5 out.npz
    python tools/keras_npz_bridge.py npz-to-h5 in.npz model_config.json model_weight.h5

h5-to-npz: Model.from_config + load_weights (agent/model.py:82-92), then one array per Keras weight under its own
name - the format reversi_alpha_zero_amd.agent.model.ReversiModel.load reads (layers are matched by kind, creation
number and shape, so any name offset works).
npz-to-h5: builds the reference's graph (agent/model.py:28-58) for the architecture the arrays imply, assigns the
arrays by the same matching, and writes get_config() JSON + save_weights() h5 (agent/model.py:94-101) - what the
reference's opt / eval workers and GUI load.  Nothing else in this repository imports this file."""
import json
import re
import sys

import numpy as np


def suffix(name):
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def h5_to_npz(config_path, weight_path, out_path):
    from keras.engine.training import Model
    with open(config_path) as f:
        model = Model.from_config(json.load(f))
    model.load_weights(weight_path)
    arrays = {}
    for layer in model.layers:
        for w, v in zip(layer.weights, layer.get_weights()):
            arrays[w.name if "/" in w.name else f"{layer.name}/{w.name}"] = v
    np.savez(out_path, **arrays)


def npz_to_h5(npz_path, config_path, weight_path):
    from types import SimpleNamespace
    from reversi_zero.agent.model import ReversiModel   # the reference package must be importable
    z = np.load(npz_path)
    layers = {}
    for key in z.files:
        lname, _, wname = key.partition("/")
        layers.setdefault(lname, {})[wname.split(":")[0].split("/")[-1]] = z[key]
    convs = sorted((n for n in layers if "kernel" in layers[n] and layers[n]["kernel"].ndim == 4), key=suffix)
    bns = sorted((n for n in layers if "moving_variance" in layers[n]), key=suffix)
    denses = [n for n in layers if "kernel" in layers[n] and layers[n]["kernel"].ndim == 2]
    v1 = next(n for n in denses if layers[n]["kernel"].shape[0] == 64 and layers[n]["kernel"].shape[1] != 1)   # 64 -> value_fc_size
    mc = SimpleNamespace(cnn_filter_num=int(layers[convs[0]]["kernel"].shape[3]), cnn_filter_size=int(layers[convs[0]]["kernel"].shape[0]),
                         res_layer_num=(len(convs) - 3) // 2, l2_reg=1e-4, value_fc_size=int(layers[v1]["kernel"].shape[1]))
    rm = ReversiModel(SimpleNamespace(model=mc))
    rm.build()
    kconvs = sorted((l for l in rm.model.layers if l.__class__.__name__ == "Conv2D"), key=lambda l: suffix(l.name))
    kbns = sorted((l for l in rm.model.layers if l.__class__.__name__ == "BatchNormalization"), key=lambda l: suffix(l.name))
    for kl, n in zip(kconvs, convs):
        kl.set_weights([layers[n]["kernel"], layers[n]["bias"]])
    for kl, n in zip(kbns, bns):
        kl.set_weights([layers[n][k] for k in ("gamma", "beta", "moving_mean", "moving_variance")])
    for kl in (l for l in rm.model.layers if l.__class__.__name__ == "Dense"):
        shape = tuple(kl.get_weights()[0].shape)
        n = next(n for n in denses if tuple(layers[n]["kernel"].shape) == shape)
        kl.set_weights([layers[n]["kernel"], layers[n]["bias"]])
    rm.save(config_path, weight_path)


if __name__ == "__main__":
    if len(sys.argv) != 5 or sys.argv[1] not in ("h5-to-npz", "npz-to-h5"):
        raise SystemExit(__doc__)
    (h5_to_npz if sys.argv[1] == "h5-to-npz" else npz_to_h5)(*sys.argv[2:])
