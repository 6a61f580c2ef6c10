"""What the three factory modules of the plugin surface share (basic_vqa/model_factory.py, basic_vqa/architect_factory.py,
darts_vqa/factory.py): how a model's constructor arguments are read from a config / args object and a dataset, and how
the optimizer and its schedule are made.  The factories themselves only keep the reference's function names."""
from torch.optim import lr_scheduler

from .optim import FusedAdam


def vocab_sizes(dataset):
    """(question vocabulary size, answer vocabulary size): all the factories need from a dataset."""
    return dataset.qst_vocab.vocab_size, dataset.ans_vocab.vocab_size


def adam(model, lr):
    """torch.optim.Adam-compatible optimizer on the HIP kernels (optim.FusedAdam; state_dict interchangeable).  The flat
    arena is built on the first ``step()`` (the driver creates its optimizers before it moves the models to the device), so
    ``optimizer.step()`` is one kernel launch for all parameters that train -- 2151 tensors in the PC-DARTS examiner."""
    return FusedAdam(model.parameters(), lr=lr, flatten='auto')


def step_decay(optimizer, step_size, gamma):
    return lr_scheduler.StepLR(optimizer, step_size=step_size, gamma=gamma)
