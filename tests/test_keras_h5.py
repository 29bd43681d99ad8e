"""Keras weight files without h5py (agent/model.py:82-101 of the reference): the reader against files
written by REAL h5py / libhdf5 (tests/golden/keras_h5/, made by tests/golden/make_golden_keras_h5.py), the writer against
the reader and - where an interpreter with h5py exists (this container: /opt/conda/bin/python3.9) - against h5py itself
running the steps of Keras' load_weights."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import make_golden_keras_h5 as gen   # noqa: E402  (numpy only at import time)

FIX = os.path.join(GOLDEN, "keras_h5")
H5PY_PYTHON = next((p for p in ("/opt/conda/bin/python3.9",) if os.path.exists(p)), None)


def _have_h5py():
    if H5PY_PYTHON is None:
        return False
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True).returncode == 0


needs_h5py = pytest.mark.skipif(not _have_h5py(), reason="no interpreter with h5py")


@pytest.mark.parametrize("fname", ["mini_fixed.h5", "mini_vlen.h5", "mini_gzip.h5"])
def test_reader_against_h5py_written_files(fname):
    from reversi_alpha_zero_amd.lib.keras_h5 import H5File, read_keras_weights
    arrays, info = read_keras_weights(os.path.join(FIX, fname))
    want = gen.expected_arrays()
    assert list(arrays) == list(want)                     # file order = Keras' layer order, weight order within a layer
    for k in want:
        assert arrays[k].dtype == np.float32 and arrays[k].shape == want[k].shape and np.array_equal(arrays[k], want[k]), k
    assert info["layer_names"] == [l for l, _ in gen.layers()] and info["backend"] == "tensorflow" and info["keras_version"] == "2.1.2"
    f = H5File(os.path.join(FIX, fname))
    assert f.keys() == sorted(l for l, _ in gen.layers())
    assert "conv2d_1/conv2d_1/kernel:0" in f and "conv2d_1/nothing" not in f
    assert f["activation_1"].keys() == [] and f["conv2d_1"]["conv2d_1"].keys() == ["bias:0", "kernel:0"]
    assert f["dense_1/dense_1/kernel:0"].shape == (64, 16)


@pytest.mark.parametrize("fname", ["mini_fixed.h5", "mini_vlen.h5"])
def test_model_load_reads_a_keras_h5(tmp_path, fname):
    """ReversiModel.load (agent/model.py:82-92) on an h5py-written file == the net assembled from the same arrays;
    the architecture (16 filters, 1 block, value_fc 16) is recovered from the file, whatever the Config says."""
    import torch
    from reversi_alpha_zero_amd.agent.model import ReversiModel, net_from_keras_named_arrays
    from reversi_alpha_zero_amd.config import Config
    cpath = str(tmp_path / "model_config.json")
    open(cpath, "w").write("{}")
    m = ReversiModel(Config())
    assert m.load(cpath, os.path.join(FIX, fname))
    net = net_from_keras_named_arrays(gen.expected_arrays())
    assert (m.model.filters, m.model.res_layers, m.model.value_fc) == (16, 1, 16)
    assert m.model.to_blob() == net.to_blob()
    assert m.digest == hashlib.sha256(open(os.path.join(FIX, fname), "rb").read()).hexdigest()
    x = torch.zeros(3, 2, 8, 8)
    x[0, 0, 3, 4] = x[0, 1, 3, 3] = x[1, 0, 0, 0] = 1
    with torch.no_grad():
        (p1, v1), (p2, v2) = m.model(x), net(x)
    assert torch.equal(p1, p2) and torch.equal(v1, v2)
    # spot values straight from Keras layouts: Conv2D kernel (kh, kw, in, out) -> torch (out, in, kh, kw); Dense (in, out) -> (out, in)
    a = gen.expected_arrays()
    assert float(m.model.stem.conv.weight.detach()[5, 1, 2, 0]) == float(a["conv2d_1/kernel:0"][2, 0, 1, 5])
    assert float(m.model.policy_conv.conv.weight.detach()[1, 7, 0, 0]) == float(a["conv2d_4/kernel:0"][0, 0, 7, 1])
    assert float(m.model.value_conv.bn.running_var[0]) == float(a["batch_normalization_5/moving_variance:0"][0])
    assert float(m.model.value_fc1.weight.detach()[3, 60]) == float(a["dense_1/kernel:0"][60, 3])


def test_save_writes_keras_files_and_round_trips(tmp_path):
    """ReversiModel.save (agent/model.py:94-101): an HDF5 weight file in Keras' save_weights layout + Keras' get_config
    JSON; loading it back gives the same blob; equal weights -> equal bytes (the digest identifies the model)."""
    from reversi_alpha_zero_amd.agent.model import ReversiModel, ReversiNet, keras_layers
    from reversi_alpha_zero_amd.config import Config
    from reversi_alpha_zero_amd.lib.keras_h5 import SIGNATURE, H5File, read_keras_weights
    net = ReversiNet(32, 2, 48).keras_init_(1).randomize_bn_(2)
    m = ReversiModel(Config())
    m.model = net
    cpath, wpath = str(tmp_path / "model_config.json"), str(tmp_path / "model_weight.h5")
    m.save(cpath, wpath)
    raw = open(wpath, "rb").read()
    assert raw[:8] == SIGNATURE and m.digest == hashlib.sha256(raw).hexdigest()
    m.save(str(tmp_path / "c2.json"), str(tmp_path / "w2.h5"))
    assert open(str(tmp_path / "w2.h5"), "rb").read() == raw          # no time stamps: content-addressed
    arrays, info = read_keras_weights(wpath)
    table = keras_layers(net)
    assert info["layer_names"] == [n for n, _, _ in table] and info["keras_version"] == "2.1.2"
    # Keras' load_weights pairs weighted layers BY ORDER: value head's conv comes before the policy head's (one level deeper)
    weighted = [n for n in info["layer_names"] if H5File(wpath)[n].attrs["weight_names"].size]
    assert weighted[-7:] == ["conv2d_7", "conv2d_6", "batch_normalization_7", "batch_normalization_6", "dense_1", "policy_out", "value_out"]
    assert arrays["conv2d_7/kernel:0"].shape == (1, 1, 32, 1) and arrays["conv2d_6/kernel:0"].shape == (1, 1, 32, 2)
    m2 = ReversiModel(Config())
    assert m2.load(cpath, wpath) and m2.model.to_blob() == net.to_blob() and m2.digest == m.digest
    # the config JSON: Keras 2.1 Model.get_config() of agent/model.py:28-58
    c = json.load(open(cpath))
    assert c["name"] == "reversi_model" and [l["name"] for l in c["layers"]] == info["layer_names"]
    assert c["input_layers"] == [["input_1", 0, 0]] and c["output_layers"] == [["policy_out", 0, 0], ["value_out", 0, 0]]
    by = {l["name"]: l for l in c["layers"]}
    assert by["input_1"]["config"]["batch_input_shape"] == [None, 2, 8, 8]
    assert by["conv2d_1"]["config"]["filters"] == 32 and by["conv2d_1"]["config"]["padding"] == "same" \
        and by["conv2d_1"]["config"]["data_format"] == "channels_first"
    assert by["conv2d_6"]["config"]["filters"] == 2 and by["conv2d_6"]["config"]["kernel_size"] == [1, 1]
    assert by["batch_normalization_3"]["config"]["axis"] == 1 and by["batch_normalization_3"]["config"]["epsilon"] == 1e-3
    assert by["add_1"]["inbound_nodes"] == [[["activation_1", 0, 0, {}], ["batch_normalization_3", 0, 0, {}]]]
    assert by["add_2"]["inbound_nodes"] == [[["activation_3", 0, 0, {}], ["batch_normalization_5", 0, 0, {}]]]
    assert by["dense_1"]["config"]["units"] == 48 and by["value_out"]["config"]["activation"] == "tanh" \
        and by["policy_out"]["config"]["activation"] == "softmax"
    # every layer's inputs are defined before it (Model.from_config processes layers in order, deferring is not needed)
    seen = set()
    for l in c["layers"]:
        assert all(i[0] in seen for node in l["inbound_nodes"] for i in node), l["name"]
        seen.add(l["name"])


def test_keras_layer_order_is_by_depth():
    """keras_layers restates Keras' depth sort: recompute depth = longest path to an output from the inbound lists and
    check the table is sorted by it (ties: policy head before value head)."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet, keras_layers
    for R in (0, 1, 3):
        table = keras_layers(ReversiNet(8, R, 8))
        consumers = {n: [] for n, _, _ in table}
        for n, _, inbound in table:
            for i in inbound:
                consumers[i].append(n)
        depth = {}

        def d(n):
            if n not in depth:
                depth[n] = 0 if not consumers[n] else 1 + max(d(c) for c in consumers[n])
            return depth[n]
        ds = [d(n) for n, _, _ in table]
        assert ds == sorted(ds, reverse=True) and len(table) == 4 + 7 * R + 11
        assert sum(cls == "Conv2D" for _, cls, _ in table) == 2 * R + 3


H5PY_CHECK = r"""
import sys, json, numpy as np, h5py
f = h5py.File(sys.argv[1], 'r')
# keras/engine/topology.py load_weights_from_hdf5_group (2.1.2), restated
kv = f.attrs['keras_version'].decode('utf8') if 'keras_version' in f.attrs else '1'
backend = f.attrs['backend'].decode('utf8') if 'backend' in f.attrs else None
layer_names = [n.decode('utf8') for n in f.attrs['layer_names']]
out, filtered = {}, []
for name in layer_names:
    g = f[name]
    weight_names = [n.decode('utf8') for n in g.attrs['weight_names']]
    if weight_names:
        filtered.append(name)
    for wn in weight_names:
        out[wn] = np.asarray(g[wn])
np.savez(sys.argv[2], **{k.replace('/', '|'): v for k, v in out.items()})
json.dump({'keras_version': kv, 'backend': backend, 'layer_names': layer_names, 'filtered': filtered,
           'dtypes': sorted({str(v.dtype) for v in out.values()})}, open(sys.argv[3], 'w'))
"""


@needs_h5py
def test_h5py_reads_what_the_writer_wrote(tmp_path):
    """The writer's file through libhdf5: h5py runs the steps of Keras' load_weights on it and finds every weight."""
    from reversi_alpha_zero_amd.agent.model import ReversiModel, ReversiNet, keras_named_arrays
    from reversi_alpha_zero_amd.config import Config
    net = ReversiNet(256, 10, 256).keras_init_(3).randomize_bn_(4)   # the metric's architecture: 101 layers, 94 MB
    m = ReversiModel(Config())
    m.model = net
    wpath = str(tmp_path / "model_best_weight.h5")
    m.save(str(tmp_path / "model_best_config.json"), wpath)
    r = subprocess.run([H5PY_PYTHON, "-c", H5PY_CHECK, wpath, str(tmp_path / "out.npz"), str(tmp_path / "out.json")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    info = json.load(open(str(tmp_path / "out.json")))
    assert info["keras_version"] == "2.1.2" and info["backend"] == "tensorflow" and info["dtypes"] == ["float32"]
    assert len(info["layer_names"]) == 4 + 70 + 11 and len(info["filtered"]) == 23 + 23 + 3
    want = keras_named_arrays(net)
    with np.load(str(tmp_path / "out.npz")) as z:
        assert {k.replace("|", "/") for k in z.files} == set(want)
        for k, v in want.items():
            assert np.array_equal(z[k.replace("/", "|")], v), k
    # and libhdf5's own checker walks the whole file without complaint
    h5ls = os.path.join(os.path.dirname(H5PY_PYTHON), "h5ls")
    if os.path.exists(h5ls):
        r = subprocess.run([h5ls, "-r", wpath], capture_output=True, text=True)
        assert r.returncode == 0 and "/conv2d_23/conv2d_23/kernel:0" in r.stdout and "Dataset {1, 1, 256, 1}" in r.stdout, r.stdout[-500:]


def test_large_groups_and_other_dtypes(tmp_path):
    """Groups beyond one symbol-table node / one B-tree node, scalar + integer + float64 data, empty arrays."""
    from reversi_alpha_zero_amd.lib.keras_h5 import H5File, write_h5
    kids = {f"layer_{i:04d}": ({"idx": np.int32(i)}, {"w": np.full((2, 3), i, np.float64), "n": np.arange(i % 5, dtype=np.int16)})
            for i in range(700)}
    path = str(tmp_path / "big.h5")
    write_h5(path, {"title": "many groups", "counts": np.arange(4, dtype=np.uint8)}, kids)
    f = H5File(path)
    assert f.attrs["title"] == b"many groups" and f.attrs["counts"].tolist() == [0, 1, 2, 3]
    assert f.keys() == sorted(kids)
    for i in (0, 7, 8, 255, 256, 257, 699):
        g = f[f"layer_{i:04d}"]
        assert int(g.attrs["idx"]) == i and g["w"].read().dtype == np.float64 and (g["w"].read() == i).all()
        assert g["n"].read().tolist() == list(range(i % 5)) and g["n"].dtype == np.int16
    if _have_h5py():
        code = ("import h5py,sys; f=h5py.File(sys.argv[1],'r'); assert len(f)==700; "
                "assert all(int(f['layer_%04d'%i].attrs['idx'])==i and (f['layer_%04d/w'%i][...]==i).all() for i in range(700)); "
                "assert f.attrs['title']==b'many groups'")
        r = subprocess.run([H5PY_PYTHON, "-c", code, path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_refusals(tmp_path):
    """Not HDF5, cut short, or outside the supported subset -> H5FormatError (a ValueError) that says why."""
    from reversi_alpha_zero_amd.lib.keras_h5 import H5FormatError, H5File, read_keras_weights, write_h5
    with pytest.raises(H5FormatError, match="signature"):
        H5File(b"PK\x03\x04" + bytes(100))
    raw = open(os.path.join(FIX, "mini_fixed.h5"), "rb").read()
    with pytest.raises(H5FormatError, match="truncated"):
        read_keras_weights(raw[:len(raw) // 2])
    v2 = bytearray(raw[:96])
    v2[8] = 2
    with pytest.raises(H5FormatError, match="libver"):
        H5File(bytes(v2))
    with pytest.raises(H5FormatError, match="layer_names"):
        read_keras_weights(write_h5(None, {"x": 1}, {"d": np.zeros(3, np.float32)}))
    assert issubclass(H5FormatError, ValueError)
    if _have_h5py():   # a real libver='latest' file: version-2 structures are named, not misread
        p = str(tmp_path / "latest.h5")
        r = subprocess.run([H5PY_PYTHON, "-c", "import h5py,sys; f=h5py.File(sys.argv[1],'w',libver='latest'); f['x']=[1.0,2.0]; f.close()", p])
        assert r.returncode == 0
        with pytest.raises(H5FormatError, match="libver|version"):
            H5File(p)["x"]


def test_model_save_style_file_is_read(tmp_path):
    """A Keras `model.save()` file keeps the same layout one level down, in the group `model_weights`: read too."""
    from reversi_alpha_zero_amd.agent.model import ReversiNet, keras_named_arrays, keras_weight_layers, net_from_keras_named_arrays
    from reversi_alpha_zero_amd.lib.keras_h5 import H5File, read_keras_weights, write_h5, write_keras_weights
    net = ReversiNet(16, 1, 16).keras_init_(9).randomize_bn_(10)
    inner = H5File(write_keras_weights(None, keras_weight_layers(net)))

    def tree(g):   # re-nest the plain weight file under /model_weights
        from reversi_alpha_zero_amd.lib.keras_h5 import Dataset
        return {k: (g[k].read() if isinstance(g[k], Dataset) else (dict(g[k].attrs), tree(g[k]))) for k in g.keys()}
    raw = write_h5(None, {"keras_version": "2.1.2", "model_config": "{}"}, {"model_weights": (dict(inner.attrs), tree(inner.root))})
    arrays, info = read_keras_weights(raw)
    want = keras_named_arrays(net)
    assert set(arrays) == set(want) and all(np.array_equal(arrays[k], want[k]) for k in want)
    assert net_from_keras_named_arrays(arrays).to_blob() == net.to_blob()


def test_writer_refuses_attributes_beyond_the_format_limit():
    """HDF5's classic object headers hold attributes of < 64 KiB (h5py fails the same way): a clear error, not a corrupt file."""
    from reversi_alpha_zero_amd.lib.keras_h5 import H5FormatError, write_h5
    with pytest.raises(H5FormatError, match="64 KiB"):
        write_h5(None, {"layer_names": np.array([b"x" * 40] * 2000, dtype="S")}, {})


H5PY_FUZZ = r"""
import sys, json, numpy as np, h5py
out_dir, seed = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(seed)
DT = ['<f4', '<f8', '>f4', '<i4', '<i8', '<u1', '<u2', '>i2', 'S7']
def rand_array(dt, shape):
    if dt.startswith('S'):
        return np.array([bytes(rng.integers(97, 123, rng.integers(0, 8)).astype(np.uint8)) for _ in range(int(np.prod(shape)))], dtype=dt).reshape(shape)
    d = np.dtype(dt)
    if d.kind == 'f':
        return rng.standard_normal(shape).astype(d)
    info = np.iinfo(d)
    return rng.integers(info.min, info.max, shape, dtype=d.newbyteorder('=')).astype(d)
manifest = {}
def fill(g, path, depth):
    for a in range(rng.integers(0, 4)):
        dt = DT[rng.integers(len(DT))]
        kind = rng.integers(3)
        if kind == 0:
            v = rand_array(dt, ())
        elif kind == 1:
            v = rand_array(dt, (int(rng.integers(0, 6)),))
        else:
            v = rand_array(dt, (int(rng.integers(1, 4)), int(rng.integers(1, 4))))
        name = 'attr%d' % a
        g.attrs[name] = v
        manifest[path + '@' + name] = np.asarray(v)
    if rng.integers(4) == 0:
        g.attrs['vlen'] = [b'alpha', b'be', b'']
        manifest[path + '@vlen'] = np.array([b'alpha', b'be', b''], dtype='S5')
    for i in range(rng.integers(1, 12 if depth == 0 else 5)):
        name = ('n%d_%s' % (i, 'x' * int(rng.integers(0, 20))))
        if depth < 2 and rng.integers(3) == 0:
            fill(g.create_group(name), path + '/' + name, depth + 1)
        else:
            dt = DT[rng.integers(len(DT))]
            nd = int(rng.integers(0, 4))
            shape = tuple(int(rng.integers(0 if nd == 1 else 1, 9)) for _ in range(nd))
            v = rand_array(dt, shape)
            kw = {}
            if nd >= 1 and int(np.prod(shape)) > 0 and not dt.startswith('S') and rng.integers(3) == 0:
                kw = dict(chunks=tuple(max(1, s // 2) for s in shape), compression='gzip' if rng.integers(2) else None,
                          shuffle=bool(rng.integers(2)), fletcher32=bool(rng.integers(2)))
            g.create_dataset(name, data=v, **kw)
            manifest[path + '/' + name] = v
with h5py.File(out_dir + '/f.h5', 'w') as f:
    fill(f, '', 0)
np.savez(out_dir + '/manifest.npz', **{k.replace('/', '|'): v for k, v in manifest.items()})
"""


@needs_h5py
def test_reader_on_random_h5py_files(tmp_path):
    """Differential test of the reader beyond the Keras layout: random group trees, dtypes (both byte orders, ints, floats,
    fixed strings), scalar / empty / 2-d attributes, variable-length string attributes, chunked + filtered datasets, all
    written by h5py; every dataset and attribute read back equal."""
    from reversi_alpha_zero_amd.lib.keras_h5 import Dataset, Group, H5File

    def walk(g, path, out):
        for k, v in g.attrs.items():
            out[path + "@" + k] = v
        for name in g.keys():
            child = g[name]
            if isinstance(child, Dataset):
                out[path + "/" + name] = child.read()
                for k, v in child.attrs.items():
                    out[path + "/" + name + "@" + k] = v
            else:
                walk(child, path + "/" + name, out)
    total = 0
    for seed in range(int(os.environ.get("RAZ_H5_FUZZ_SEEDS", "12"))):
        d = tmp_path / f"s{seed}"
        d.mkdir()
        r = subprocess.run([H5PY_PYTHON, "-c", H5PY_FUZZ, str(d), str(seed)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        got = {}
        walk(H5File(str(d / "f.h5")).root, "", got)
        with np.load(str(d / "manifest.npz")) as z:
            want = {k.replace("|", "/"): z[k] for k in z.files}
        assert set(got) == set(want), (seed, sorted(set(got) ^ set(want))[:5])
        for k, w in want.items():
            g = got[k]
            if isinstance(g, list):   # variable-length strings
                g = np.array(g, dtype="S5")
            g = np.asarray(g)
            assert g.shape == w.shape and g.dtype.kind == w.dtype.kind, (seed, k, g.dtype, w.dtype, g.shape, w.shape)
            assert g.dtype.kind == "S" or g.dtype.itemsize == w.dtype.itemsize, (seed, k, g.dtype, w.dtype)   # (numpy trims scalar strings)
            assert np.array_equal(g, w), (seed, k)
            total += 1
    assert total > 150


H5PY_DUMP = r"""
import sys, numpy as np, h5py
out = {}
def visit(name, obj):
    for k, v in obj.attrs.items():
        out[name + '@' + k] = np.asarray(v)
    if isinstance(obj, h5py.Dataset):
        out[name] = np.asarray(obj[...])
with h5py.File(sys.argv[1], 'r') as f:
    for k, v in f.attrs.items():
        out['@' + k] = np.asarray(v)
    f.visititems(visit)
np.savez(sys.argv[2], **{k.replace('/', '|'): v for k, v in out.items()})
"""


@needs_h5py
def test_writer_on_random_trees_read_by_h5py(tmp_path):
    """The other direction: random trees (nested groups of any size, float / int arrays of rank 0..3 incl. empty ones,
    string and numeric attributes) written by write_h5, read back through h5py / libhdf5."""
    from reversi_alpha_zero_amd.lib.keras_h5 import write_h5
    total = 0
    for seed in range(int(os.environ.get("RAZ_H5_FUZZ_SEEDS", "8"))):
        rng = np.random.default_rng(1000 + seed)
        want = {}

        def rand(nd=None):
            dt = ["<f4", "<f8", "<i4", "<i8", "<u1", "<i2"][rng.integers(6)]
            nd = int(rng.integers(0, 4)) if nd is None else nd
            shape = tuple(int(rng.integers(0 if nd == 1 else 1, 7)) for _ in range(nd))
            if dt[1] == "f":
                return rng.standard_normal(shape).astype(dt)
            info = np.iinfo(np.dtype(dt))
            return rng.integers(info.min, info.max, shape, dtype=np.dtype(dt))

        def attrs(path):
            a = {}
            for i in range(rng.integers(0, 4)):
                k = f"a{i}"
                kind = rng.integers(3)
                a[k] = rand() if kind == 0 else ("text %d" % rng.integers(1000) if kind == 1 else
                                                 np.array([b"n%d" % j for j in range(rng.integers(1, 5))], dtype="S"))
                want[path + "@" + k] = np.asarray(a[k].encode() if isinstance(a[k], str) else a[k])
            return a

        def tree(path, depth):
            kids = {}
            for i in range(rng.integers(0, 40 if depth == 0 else 6)):
                name = f"k{i}_" + "y" * int(rng.integers(0, 12))
                if depth < 2 and rng.integers(4) == 0:
                    p = (path + "/" + name).lstrip("/")
                    a = attrs(p)
                    kids[name] = (a, tree(p, depth + 1))
                else:
                    kids[name] = rand()
                    want[(path + "/" + name).lstrip("/")] = kids[name]
            return kids
        root_attrs = attrs("")
        path = str(tmp_path / f"w{seed}.h5")
        write_h5(path, root_attrs, tree("", 0))
        r = subprocess.run([H5PY_PYTHON, "-c", H5PY_DUMP, path, str(tmp_path / f"w{seed}.npz")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        with np.load(str(tmp_path / f"w{seed}.npz")) as z:
            got = {k.replace("|", "/"): z[k] for k in z.files}
        assert set(got) == set(want), (seed, sorted(set(got) ^ set(want))[:6])
        for k, w in want.items():
            g = got[k]
            assert g.shape == w.shape and g.dtype.kind == w.dtype.kind and np.array_equal(g, w), (seed, k, g.dtype, w.dtype)
            total += 1
    assert total > 100
