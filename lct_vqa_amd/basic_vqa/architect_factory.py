"""``get_architect`` of the basic_vqa plugin surface (reference basic_vqa/architect_factory.py:5-16): which architecture
search runs beside the weight training, chosen by the driver's ``config.ARCH_TYPE`` / ``config.SKIP_STAGE2``."""
from . import config as _config
from .pcdarts.architect import Architect
from .pcdarts.architect_lct import ArchitectLct


def get_architect(ef_model, w_model, ef_optimizer, w_optimizer):
    cfg = _config.active()
    if cfg.ARCH_TYPE == 'fixed':                      # hand-designed encoder: nothing to search
        return None
    assert cfg.ARCH_TYPE == 'darts', f'unrecognized ARCH_TYPE {cfg.ARCH_TYPE!r}'
    if cfg.SKIP_STAGE2:                               # no student model in the loop: plain 2-level DARTS on the examiner
        return Architect(ef_model)
    return ArchitectLct(ef_model, w_model, ef_optimizer, w_optimizer)
