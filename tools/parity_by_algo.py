#!/usr/bin/env python3
"""Logit error of the full-size W model (BASELINE configs[1] shapes, B=4) against the reference's golden logits, per trunk
convolution algorithm.  Needs tests/golden/w_full_b4.pt (made by tests/make_golden.py from the reference's own modules)."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _util import load, rel_err
from lct_vqa_amd import ops
from lct_vqa_amd.basic_vqa import models, config
config.DEVICE = torch.device('cuda')
fx = load('w_full_b4.pt')
for algo in ('direct', 'winograd2', 'winograd'):
    ops.set_conv_algo(algo)
    torch.manual_seed(fx['seed'])
    m = models.VqaModel(**fx['dims']).to('cuda').eval()
    torch.manual_seed(fx['data_seed'])
    img = torch.randn(fx['B'], 3, 224, 224).to('cuda')
    with torch.no_grad():
        lg = m(img, fx['question'].to('cuda'))
    top2 = fx['eval_logits'].topk(2, dim=1).values
    print(f'{algo:10s} logits max-rel err {rel_err(lg, fx["eval_logits"]):.3e}   argmax identical {torch.equal(lg.argmax(1).cpu(), fx["argmax"])}'
          f'   (smallest top-1/top-2 gap / max|logit| = {((top2[:, 0] - top2[:, 1]).min() / fx["eval_logits"].abs().max()).item():.2e})')
