"""``get_architect`` of the basic_vqa plugin surface (reference basic_vqa/architect_factory.py:5-16): which architecture
search runs beside the weight training.  The choice is a table keyed by the driver's ``config.ARCH_TYPE`` and
``config.SKIP_STAGE2``:

    ARCH_TYPE   SKIP_STAGE2   architect
    'fixed'     any           None (hand-designed encoder, nothing to search)
    'darts'     True          Architect(ef_model)            2-level DARTS on the examiner alone
    'darts'     False         ArchitectLct(ef, w, opts...)   3-level learning-by-creating-tests
"""
from . import config as _config
from .pcdarts.architect import Architect
from .pcdarts.architect_lct import ArchitectLct

_SEARCH = {
    True: lambda ef_model, w_model, ef_optimizer, w_optimizer: Architect(ef_model),
    False: ArchitectLct,
}


def get_architect(ef_model, w_model, ef_optimizer, w_optimizer):
    cfg = _config.active()
    if cfg.ARCH_TYPE == 'fixed':
        return None
    if cfg.ARCH_TYPE != 'darts':
        raise AssertionError(f'unrecognized ARCH_TYPE {cfg.ARCH_TYPE!r}')
    return _SEARCH[bool(cfg.SKIP_STAGE2)](ef_model, w_model, ef_optimizer, w_optimizer)
