"""The darts_vqa plugin surface (reference darts_vqa/factory.py:6-35): model / optimizer / scheduler / architect from the
tree's argparse namespace.  Fields read: unified, embed_size, word_embed_size, num_layers, hidden_size, arch_type,
learn_rate, step_size, gamma (+ arch_learn_rate, arch_wt_decay, qst_only inside the architect)."""
from .. import plugin
from .pcdarts.architect_vqa import Architect
from .vqa_model import VqaModel, VqaModelUnified


def get_vqa_model(args, dataset):
    shape = (args.word_embed_size, args.num_layers, args.hidden_size, args.arch_type)
    if args.unified:                                  # one vocabulary for questions and answers
        return VqaModelUnified(args.embed_size, dataset.unified_vocab.vocab_size, *shape)
    return VqaModel(args.embed_size, *plugin.vocab_sizes(dataset), *shape)


def get_optimizer(args, model):
    return plugin.adam(model, args.learn_rate)


def get_scheduler(args, optimizer):
    return plugin.step_decay(optimizer, args.step_size, args.gamma)


def get_architect(args, model):
    if args.arch_type == 'vgg':
        return None
    if args.arch_type == 'darts':
        return Architect(model, args)
    raise Exception(f'Unrecognized arch_type: {args.arch_type}')
