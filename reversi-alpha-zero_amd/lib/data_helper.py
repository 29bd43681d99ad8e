"""play_*.json file helpers — the wire format between the `self` worker and the reference's `opt`
worker (reversi_zero/lib/data_helper.py:11-30, consumer worker/optimize.py:165-231).  A file is one
JSON list of rows  [[own:int, enemy:int], [64 floats], z:int]  written with json.dump defaults."""
import json
import os
from glob import glob


def get_game_data_filenames(rc):
    pattern = os.path.join(rc.play_data_dir, rc.play_data_filename_tmpl % "*")
    return list(sorted(glob(pattern)))


def get_next_generation_model_dirs(rc):
    """lib/data_helper.py:17-20."""
    dir_pattern = os.path.join(rc.next_generation_model_dir, rc.next_generation_model_dirname_tmpl % "*")
    return list(sorted(glob(dir_pattern)))


def write_game_data_to_file(path, data):
    with open(path, "wt") as f:
        json.dump(data, f)


def read_game_data_from_file(path):
    with open(path, "rt") as f:
        return json.load(f)


# ---- the `opt` worker's view of the rows (worker/optimize.py:214-231) ----------------------------------
def pack_game_data(data):
    """Rows [[own, enemy], policy64, z] -> compact arrays (own u64[N], enemy u64[N], policy f32[N,64], z i8[N]):
    8 + 8 + 256 + 1 bytes per row instead of the JSON text's ~1 KB."""
    import numpy as np
    n = len(data)
    own = np.fromiter((row[0][0] for row in data), dtype=np.uint64, count=n)
    enemy = np.fromiter((row[0][1] for row in data), dtype=np.uint64, count=n)
    policy = np.asarray([row[1] for row in data], dtype=np.float32).reshape(n, 64)
    z = np.fromiter((row[2] for row in data), dtype=np.int8, count=n)
    return own, enemy, policy, z


def convert_to_training_data(data):
    """OptimizeWorker.convert_to_training_data (worker/optimize.py:214-231), same return value:
    (state uint8 (N,2,8,8), policy float64 (N,64), z int (N,)); the planes come from the packed
    bitboards with one vectorised shift instead of a Python bit_to_array per board."""
    import numpy as np
    own, enemy, _, _ = pack_game_data(data)
    sh = np.arange(64, dtype=np.uint64)
    state = np.stack([((own[:, None] >> sh) & np.uint64(1)), ((enemy[:, None] >> sh) & np.uint64(1))], axis=1)
    return (state.astype(np.uint8).reshape(len(data), 2, 8, 8), np.array([row[1] for row in data]),
            np.array([row[2] for row in data]))


def training_tensors(data, device="cuda:0"):
    """The same training batch resident on the GPU: the packed bitboards are uploaded (16 B/row) and
    expanded to float32 planes by `raz_planes_batch` (include/raz.h) — the on-device bit_to_array of
    SURVEY §8(f) 4.  Returns (state f32 (N,2,8,8), policy f32 (N,64), z f32 (N,)) torch tensors."""
    import numpy as np
    import torch
    from . import bitboard as bb
    own, enemy, policy, z = pack_game_data(data)
    dev = torch.device(device)
    o = torch.from_numpy(own.view(np.int64)).to(dev)
    e = torch.from_numpy(enemy.view(np.int64)).to(dev)
    planes = bb.planes_batch(o, e)
    return (planes.view(len(data), 2, 8, 8), torch.from_numpy(policy).to(dev),
            torch.from_numpy(z.astype(np.float32)).to(dev))
