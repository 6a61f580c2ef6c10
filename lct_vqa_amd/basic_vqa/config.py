"""Configuration of the basic_vqa tree: module-level names and default values are the reference's (basic_vqa/config.py:8-98),
because its driver and its users address them as ``config.NAME``.  The models of this package read ``DEVICE``,
``ARCH_TYPE``, ``W_LAMBDA`` and ``TEMPERATURE`` at CALL time through ``active()``: when the reference's own driver is in
charge (its ``config`` module is imported as top-level ``config``), ``active()`` returns THAT module, so the driver's
``update_config(args)`` is honoured by these models without touching the driver.
"""
import sys

import torch

_DEFAULTS = {
    # data
    'INPUT_DIR': '../../data/vqa/inputs64', 'MAX_QST_LEN': 30, 'MAX_NUM_ANS': 10, 'NUM_WORKERS': 8,
    'USE_OLD_DATALOADER': False, 'TRAIN_PORTION': 1.,
    # model geometry
    'IMG_EMBED_SIZE': 512, 'WORD_EMBED_SIZE': 300, 'LSTM_NUM_LAYERS': 1, 'LSTM_HIDDEN_SIZE': 512, 'PRETRAIN_ENC': True,
    # weight optimisation
    'LEARNING_RATE': 0.001, 'MOMENTUM': 0.99, 'WEIGHT_DECAY': 0, 'STEP_SIZE': 10, 'LR_DECAY': 0.1, 'GRAD_CLIP': 5,
    'BATCH_SIZE': 64, 'NUM_EPOCHS': 30,
    # architecture search / learning by creating tests
    'ARCH_TYPE': 'darts', 'ARCH_LEARNING_RATE': 6e-4, 'ARCH_WEIGHT_DECAY': 1e-3, 'ARCH_UPDATE_FREQ': 2000,
    'ARCH_UPDATE_FREQ_MIN': 100, 'ARCH_FREQ_DECAY': 0.5, 'SKIP_STAGE2': False, 'SKIP_STAGE3': True, 'TEMPERATURE': 0.1,
    'W_LAMBDA': 1.0,
    # run bookkeeping
    'EXP_NAME': 'default_exp', 'RESUME': False, 'SEED': 10, 'ROOT_STATS_DIR': './experiment_data',
}
globals().update(_DEFAULTS)
DEVICE = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def _report_freq(arch_type):
    return 10 if arch_type == 'darts' else 100


REPORT_FREQ = _report_freq(_DEFAULTS['ARCH_TYPE'])

# argparse field of the reference's driver -> configuration name it overrides (config.py:78-98)
_FROM_ARGS = {
    'batch_size': 'BATCH_SIZE', 'num_epochs': 'NUM_EPOCHS', 'train_portion': 'TRAIN_PORTION', 'exp': 'EXP_NAME',
    'resume': 'RESUME', 'num_workers': 'NUM_WORKERS', 'arch_type': 'ARCH_TYPE', 'skip_stage2': 'SKIP_STAGE2',
    'arch_update_freq': 'ARCH_UPDATE_FREQ', 'w_lambda': 'W_LAMBDA', 'skip_stage3': 'SKIP_STAGE3',
    'use_old_dataloader': 'USE_OLD_DATALOADER', 'input_dir': 'INPUT_DIR',
}


def update_config(args):
    """Apply the driver's command line, like the reference's ``update_config``."""
    g = globals()
    for field, name in _FROM_ARGS.items():
        g[name] = getattr(args, field)
    g['PRETRAIN_ENC'] = not args.no_pretrain_enc
    g['REPORT_FREQ'] = _report_freq(args.arch_type)


def active():
    """The configuration module the models should read: the driver's top-level ``config`` when it looks like one."""
    top = sys.modules.get('config')
    if top is not None and all(hasattr(top, n) for n in ('ARCH_TYPE', 'W_LAMBDA', 'DEVICE')):
        return top
    return sys.modules[__name__]
