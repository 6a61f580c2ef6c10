"""Plugin surface of the reference (basic_vqa/model_factory.py:7-48): same function names and
meaning; ``dataset`` only needs ``.qst_vocab.vocab_size`` and ``.ans_vocab.vocab_size``."""
from torch.optim import lr_scheduler

from ..optim import FusedAdam
from . import config as _config
from .models import VqaModel
from .models_lct import VqaModel as LctVqaModel


def get_ef_model(dataset):
    config = _config.active()
    return LctVqaModel(
        embed_size=config.IMG_EMBED_SIZE,
        qst_vocab_size=dataset.qst_vocab.vocab_size,
        ans_vocab_size=dataset.ans_vocab.vocab_size,
        word_embed_size=config.WORD_EMBED_SIZE,
        num_layers=config.LSTM_NUM_LAYERS,
        hidden_size=config.LSTM_HIDDEN_SIZE,
        pretrained=config.PRETRAIN_ENC)


def get_ef_optimizer(model):
    return FusedAdam(model.parameters(), lr=_config.active().LEARNING_RATE)


def get_ef_scheduler(optimizer):
    config = _config.active()
    return lr_scheduler.StepLR(optimizer, step_size=config.STEP_SIZE, gamma=config.LR_DECAY)


def get_w_model(dataset):
    config = _config.active()
    return VqaModel(
        embed_size=config.IMG_EMBED_SIZE,
        qst_vocab_size=dataset.qst_vocab.vocab_size,
        ans_vocab_size=dataset.ans_vocab.vocab_size,
        word_embed_size=config.WORD_EMBED_SIZE,
        num_layers=config.LSTM_NUM_LAYERS,
        hidden_size=config.LSTM_HIDDEN_SIZE)


def get_w_optimizer(model):
    return FusedAdam(model.parameters(), lr=_config.active().LEARNING_RATE)


def get_w_scheduler(optimizer):
    config = _config.active()
    return lr_scheduler.StepLR(optimizer, step_size=config.STEP_SIZE, gamma=config.LR_DECAY)
