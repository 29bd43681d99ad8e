"""Best-model / next-generation bookkeeping - drop-in for reversi_zero/lib/model_helpler.py (the reference's
spelling): which weight file a worker serves, and when it re-reads it (digest of the file, agent/model.py:74-80)."""
import os
from logging import getLogger
from time import sleep

from .data_helper import get_next_generation_model_dirs

logger = getLogger(__name__)


def load_best_model_weight(model, clear_session=False):
    """model_helpler.py:11-21 (clear_session is a Keras notion; ignored)."""
    rc = model.config.resource
    return model.load(rc.model_best_config_path, rc.model_best_weight_path)


def save_as_best_model(model):
    """model_helpler.py:24-30."""
    rc = model.config.resource
    os.makedirs(rc.model_dir, exist_ok=True)
    return model.save(rc.model_best_config_path, rc.model_best_weight_path)


def reload_best_model_weight_if_changed(model, clear_session=False):
    """model_helpler.py:33-47: reload when the weight file's sha256 differs from the loaded one."""
    digest = model.fetch_digest(model.config.resource.model_best_weight_path)
    if digest != model.digest:
        return load_best_model_weight(model, clear_session=clear_session)
    return False


def reload_newest_next_generation_model_if_changed(model, clear_session=False, retries=5, retry_sleep=3.0):
    """model_helpler.py:50-80: the last directory of model/next_generation/model_* (the `opt` worker adds one per
    checkpoint); loaded when its weight file exists and its digest differs from the loaded one.  The optimizer may
    still be writing: a failed load is retried, then the error is raised like the reference does."""
    rc = model.config.resource
    dirs = get_next_generation_model_dirs(rc)
    if not dirs:
        return False
    config_path = os.path.join(dirs[-1], rc.next_generation_model_config_filename)
    weight_path = os.path.join(dirs[-1], rc.next_generation_model_weight_filename)
    digest = model.fetch_digest(weight_path)
    if not digest or digest == model.digest:
        return False
    last = None
    for attempt in range(retries):
        try:
            return model.load(config_path, weight_path)
        except Exception as e:
            last = e
            logger.warning(f"error in load model: {e}")
            if attempt + 1 < retries:
                sleep(retry_sleep)
    raise RuntimeError(f"Cannot Load Model! ({last})")
