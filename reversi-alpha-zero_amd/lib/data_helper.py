"""play_*.json file helpers — the wire format between the `self` worker and the reference's `opt`
worker (reversi_zero/lib/data_helper.py:11-30, consumer worker/optimize.py:165-231).  A file is one
JSON list of rows  [[own:int, enemy:int], [64 floats], z:int]  written with json.dump defaults."""
import json
import os
from glob import glob


def get_game_data_filenames(rc):
    pattern = os.path.join(rc.play_data_dir, rc.play_data_filename_tmpl % "*")
    return list(sorted(glob(pattern)))


def write_game_data_to_file(path, data):
    with open(path, "wt") as f:
        json.dump(data, f)


def read_game_data_from_file(path):
    with open(path, "rt") as f:
        return json.load(f)
