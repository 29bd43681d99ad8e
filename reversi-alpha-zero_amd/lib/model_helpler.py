"""Best-model bookkeeping — drop-in for reversi_zero/lib/model_helpler.py (the reference's spelling)."""
from logging import getLogger

logger = getLogger(__name__)


def load_best_model_weight(model, clear_session=False):
    """model_helpler.py:11-21 (clear_session is a Keras notion; ignored)."""
    return model.load(model.config.resource.model_best_config_path, model.config.resource.model_best_weight_path)


def save_as_best_model(model):
    """model_helpler.py:24-30."""
    return model.save(model.config.resource.model_best_config_path, model.config.resource.model_best_weight_path)


def reload_best_model_weight_if_changed(model, clear_session=False):
    """model_helpler.py:33-47: reload when the weight file's sha256 differs from the loaded one."""
    logger.debug("start reload the best model if changed")
    digest = model.fetch_digest(model.config.resource.model_best_weight_path)
    if digest != model.digest:
        return load_best_model_weight(model, clear_session=clear_session)
    logger.debug("the best model is not changed")
    return False
