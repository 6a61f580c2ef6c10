"""Configuration globals with the reference's names and defaults (reference
basic_vqa/config.py:8-98).  Models read DEVICE / ARCH_TYPE / W_LAMBDA / TEMPERATURE at call
time, exactly like the reference.  When the reference's own driver is in charge (its
``config`` module is already imported as top-level ``config``), ``active()`` returns THAT
module so ``update_config(args)`` of the driver is honoured by these models.
"""
import sys

import torch

INPUT_DIR = '../../data/vqa/inputs64'
MAX_QST_LEN = 30
MAX_NUM_ANS = 10
IMG_EMBED_SIZE = 512
WORD_EMBED_SIZE = 300
LSTM_NUM_LAYERS = 1
LSTM_HIDDEN_SIZE = 512
LEARNING_RATE = 0.001
MOMENTUM = 0.99
WEIGHT_DECAY = 0
STEP_SIZE = 10
LR_DECAY = 0.1
NUM_WORKERS = 8
ARCH_LEARNING_RATE = 6e-4
ARCH_WEIGHT_DECAY = 1e-3
GRAD_CLIP = 5
TEMPERATURE = 0.1
BATCH_SIZE = 64
NUM_EPOCHS = 30
TRAIN_PORTION = 1.
EXP_NAME = 'default_exp'
RESUME = False
SEED = 10
ROOT_STATS_DIR = './experiment_data'
DEVICE = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
ARCH_TYPE = 'darts'
REPORT_FREQ = 10 if ARCH_TYPE == 'darts' else 100
ARCH_UPDATE_FREQ = 2000
ARCH_UPDATE_FREQ_MIN = 100
ARCH_FREQ_DECAY = 0.5
SKIP_STAGE2 = False
SKIP_STAGE3 = True
PRETRAIN_ENC = True
W_LAMBDA = 1.0
USE_OLD_DATALOADER = False


def update_config(args):
    """Same argparse fields as the reference's update_config (config.py:78-98)."""
    g = globals()
    g['BATCH_SIZE'] = args.batch_size
    g['NUM_EPOCHS'] = args.num_epochs
    g['TRAIN_PORTION'] = args.train_portion
    g['EXP_NAME'] = args.exp
    g['RESUME'] = args.resume
    g['NUM_WORKERS'] = args.num_workers
    g['ARCH_TYPE'] = args.arch_type
    g['SKIP_STAGE2'] = args.skip_stage2
    g['ARCH_UPDATE_FREQ'] = args.arch_update_freq
    g['PRETRAIN_ENC'] = not args.no_pretrain_enc
    g['REPORT_FREQ'] = 10 if args.arch_type == 'darts' else 100
    g['W_LAMBDA'] = args.w_lambda
    g['SKIP_STAGE3'] = args.skip_stage3
    g['USE_OLD_DATALOADER'] = args.use_old_dataloader
    g['INPUT_DIR'] = args.input_dir


def active():
    top = sys.modules.get('config')
    if top is not None and hasattr(top, 'ARCH_TYPE') and hasattr(top, 'W_LAMBDA') and hasattr(top, 'DEVICE'):
        return top
    return sys.modules[__name__]
