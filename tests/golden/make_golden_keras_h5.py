#!/opt/conda/bin/python3.9
"""tests/golden/make_golden_keras_h5.py - Keras weight files written by REAL h5py / libhdf5, as fixtures for the h5py-free
reader in reversi_alpha_zero_amd/lib/keras_h5.py.

Run with an interpreter that has h5py (in this container: /opt/conda/bin/python3.9, h5py 3.3.0 on libhdf5 1.10.6):

    /opt/conda/bin/python3.9 tests/golden/make_golden_keras_h5.py

Keras itself is not available, so the script restates what Keras 2.1.2 `save_weights` does with h5py
(keras/engine/topology.py save_weights_to_hdf5_group): root attrs layer_names / backend / keras_version, one group per
layer with attr weight_names, one float32 dataset per weight under the weight's (slash-containing) name.  It writes the
reference's mini architecture (config/mini.yml:3-8: 16 filters, 1 residual block, value_fc 16) in three flavours:

  mini_fixed.h5   names as fixed-length byte strings  (what h5py 2.x - the reference's era - made of a list of bytes)
  mini_vlen.h5    names as variable-length strings    (what h5py 3.x makes of the same assignment)
  mini_gzip.h5    as fixed, datasets chunked + shuffle + gzip + fletcher32 (a file that went through h5repack)

Weight VALUES are an integer formula (`values`) so that the test recomputes them instead of storing them twice."""
import os
import sys

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "keras_h5")
F, R, V = 16, 1, 16


def values(shape, salt, positive=False):
    """float32 array, exactly reproducible: ((i + salt) * 2654435761 mod 2^32) / 2^32 - 0.5  (|.| + 0.5 when positive)"""
    n = int(np.prod(shape))
    x = ((np.arange(n, dtype=np.uint64) + np.uint64(salt)) * np.uint64(2654435761) % np.uint64(2 ** 32)).astype(np.float64) / 2 ** 32 - 0.5
    if positive:
        x = np.abs(x) + 0.5
    return x.astype(np.float32).reshape(shape)


def layers():
    """[(layer name, [(weight name, shape)])] in Keras' model.layers order for the mini architecture (the order
    reversi_alpha_zero_amd.agent.model.keras_layers derives; the reader must not depend on it)."""
    def conv(i, cin, cout, k):
        return (f"conv2d_{i}", [(f"conv2d_{i}/kernel:0", (k, k, cin, cout)), (f"conv2d_{i}/bias:0", (cout,))])

    def bn(i, c):
        return (f"batch_normalization_{i}", [(f"batch_normalization_{i}/{w}:0", (c,)) for w in ("gamma", "beta", "moving_mean", "moving_variance")])

    def plain(name):
        return (name, [])

    def dense(name, a, b):
        return (name, [(f"{name}/kernel:0", (a, b)), (f"{name}/bias:0", (b,))])
    return [plain("input_1"), conv(1, 2, F, 3), bn(1, F), plain("activation_1"),
            conv(2, F, F, 3), bn(2, F), plain("activation_2"), conv(3, F, F, 3), bn(3, F), plain("add_1"), plain("activation_3"),
            conv(5, F, 1, 1), conv(4, F, 2, 1), bn(5, 1), bn(4, 2), plain("activation_5"), plain("activation_4"),
            plain("flatten_2"), plain("flatten_1"), dense("dense_1", 64, V), dense("policy_out", 128, 64), dense("value_out", V, 1)]


def expected_arrays():
    out, salt = {}, 1
    for _, ws in layers():
        for wname, shape in ws:
            out[wname] = values(shape, salt, positive=wname.endswith("moving_variance:0"))
            salt += 1000003
    return out


def main():
    import h5py
    os.makedirs(HERE, exist_ok=True)
    arrays = expected_arrays()
    for fname, fixed, kw in (("mini_fixed.h5", True, {}), ("mini_vlen.h5", False, {}),
                             ("mini_gzip.h5", True, dict(chunks=True, compression="gzip", shuffle=True, fletcher32=True))):
        def names(lst):
            lst = [n.encode("utf8") for n in lst]
            if not fixed:
                return lst                                   # h5py 3: list of bytes -> variable-length strings
            return np.array(lst, dtype="S") if lst else np.array([], dtype=np.float64)   # h5py 2: numpy.asarray(list)
        with h5py.File(os.path.join(HERE, fname), "w") as f:
            f.attrs["layer_names"] = names([l for l, _ in layers()])
            f.attrs["backend"] = np.bytes_(b"tensorflow") if fixed else b"tensorflow"
            f.attrs["keras_version"] = np.bytes_(b"2.1.2") if fixed else b"2.1.2"
            for lname, ws in layers():
                g = f.create_group(lname)
                g.attrs["weight_names"] = names([w for w, _ in ws])
                for wname, shape in ws:
                    d = g.create_dataset(wname, shape, dtype="float32", **kw)
                    d[...] = arrays[wname]
        print(fname, os.path.getsize(os.path.join(HERE, fname)), "bytes; h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)


if __name__ == "__main__":
    sys.exit(main())
