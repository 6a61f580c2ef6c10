#!/usr/bin/env python3
"""Uninitialised-read hunt: every torch.empty / empty_like / new_empty made while the models run is filled with NaN
(floating point) first, so a kernel that reads an element no kernel wrote turns the outputs NaN.
    python tools/poison_check.py [darts|fixed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float('nan'))
    return t


def main(kind):
    from lct_vqa_amd import ops, bilevel, darts_ops
    from lct_vqa_amd._lib import lib
    from lct_vqa_amd.basic_vqa import config, models_lct
    lib.load()
    config.DEVICE = torch.device('cuda')
    config.ARCH_TYPE = kind
    torch.manual_seed(10)
    dims = dict(embed_size=32, qst_vocab_size=50, ans_vocab_size=40, word_embed_size=16, num_layers=1, hidden_size=32)
    ef = models_lct.VqaModel(**dims, **({} if kind == 'darts' else {'pretrained': False})).to('cuda').train()
    torch.manual_seed(30)
    img = torch.randn(2, 3, 32, 32, device='cuda')
    qst = torch.randint(50, (2, 8), device='cuda'); qst[:, 0] = 2
    lab = torch.randint(40, (2,), device='cuda')
    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
    # the scratch pool is uninitialised by contract (producers write what consumers read): poison it as well
    darts_ops._scratch_pool.clear()
    ops._ws_cache.clear()
    inner_scratch, inner_ws = darts_ops._scratch, ops._workspace
    darts_ops._scratch = lambda *a: inner_scratch(*a).fill_(float('nan'))          # re-poisoned at EVERY hand-out
    ops._workspace = darts_ops._workspace = lambda *a: inner_ws(*a).fill_(float('nan'))
    with ops.dropout_disabled():
        loss = ef._loss(img, qst, lab)
        targets = (list(ef.arch_parameters()) if kind == 'darts' else []) + list(ef.parameters())
        g = torch.autograd.grad(loss, targets, allow_unused=True)
    names = (['alpha_n', 'alpha_r', 'beta_n', 'beta_r'] if kind == 'darts' else []) + [n for n, _ in ef.named_parameters()]
    bad = [n for n, x in zip(names, g) if x is not None and not torch.isfinite(x).all()]
    print('loss', float(loss), 'non-finite gradients:', len(bad), bad[:20])
    q, a = ef.generate(img)
    print('generate finite', bool(torch.isfinite(a).all()))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'darts')
