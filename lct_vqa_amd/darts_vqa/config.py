"""Reference darts_vqa/config.py:1-8.  ``active()`` returns the driver's own top-level ``config`` when the
reference's darts_vqa driver is in charge."""
import sys

import torch

DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'
SEED = 10
ROOT_STATS_DIR = './experiment_data'


def active():
    top = sys.modules.get('config')
    if top is not None and hasattr(top, 'DEVICE') and hasattr(top, 'ROOT_STATS_DIR') and not hasattr(top, 'ARCH_TYPE'):
        return top
    return sys.modules[__name__]
