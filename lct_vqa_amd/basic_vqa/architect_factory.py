"""Reference basic_vqa/architect_factory.py:5-16: None for ARCH_TYPE 'fixed'; the PC-DARTS
architects ('darts') arrive with the DARTS encoder kernels."""
from . import config as _config


def get_architect(ef_model, w_model, ef_optimizer, w_optimizer):
    config = _config.active()
    if config.ARCH_TYPE == 'fixed':
        return None
    elif config.ARCH_TYPE == 'darts':
        raise NotImplementedError(
            "lct_vqa_amd: ARCH_TYPE='darts' (PC-DARTS encoder + Architect/ArchitectLct) is not built yet")
    else:
        assert False and 'unrecognized ARCH_TYPE'
