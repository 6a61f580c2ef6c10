"""The basic_vqa plugin surface: the six functions reference basic_vqa/model_factory.py:7-48 exports and
``experiment.py`` imports with ``*``.  'ef' is the LCT examiner (models_lct.VqaModel), 'w' the student (models.VqaModel);
``dataset`` only needs ``.qst_vocab.vocab_size`` and ``.ans_vocab.vocab_size``.  Hyper-parameters come from the driver's
``config`` module (``config.active()``)."""
from .. import plugin
from . import config as _config
from .models import VqaModel
from .models_lct import VqaModel as LctVqaModel


def _dims(dataset):
    cfg = _config.active()
    qv, av = plugin.vocab_sizes(dataset)
    return dict(embed_size=cfg.IMG_EMBED_SIZE, qst_vocab_size=qv, ans_vocab_size=av, word_embed_size=cfg.WORD_EMBED_SIZE,
                num_layers=cfg.LSTM_NUM_LAYERS, hidden_size=cfg.LSTM_HIDDEN_SIZE)


def _optimizer(model):
    return plugin.adam(model, _config.active().LEARNING_RATE)


def _scheduler(optimizer):
    cfg = _config.active()
    return plugin.step_decay(optimizer, cfg.STEP_SIZE, cfg.LR_DECAY)


def get_ef_model(dataset):
    return LctVqaModel(pretrained=_config.active().PRETRAIN_ENC, **_dims(dataset))


def get_w_model(dataset):
    return VqaModel(**_dims(dataset))


get_ef_optimizer = get_w_optimizer = _optimizer
get_ef_scheduler = get_w_scheduler = _scheduler
