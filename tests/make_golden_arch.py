#!/usr/bin/env python3
"""Golden vectors of the DARTS-encoder models and of the architect steps, produced by RUNNING the reference's own
classes on CPU.   python tests/make_golden_arch.py basic|darts
  basic: basic_vqa models_lct.VqaModel(ARCH_TYPE='darts') forward/loss/grads + ArchitectLct.step (3-level)
  darts: darts_vqa vqa_model.VqaModel('darts') forward/loss/grads + pcdarts.architect_vqa.Architect.step
nn.Dropout is neutralised for this run by patching torch.nn.functional.dropout to the identity (the architects build
fresh models with ``new()`` whose Dropout holders are in training mode, and the CPU Bernoulli stream cannot be replayed
on the GPU); the product side runs under ``ops.dropout_disabled()``.  BatchNorm stays in training mode (batch statistics).  torchvision.models.vgg19 = oracle/tv_vgg19.py (see tests/make_golden.py)."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tv_vgg19  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def stub_tv():
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvm.vgg19 = tv_vgg19.vgg19
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    return m


def small(sd):
    return {k: v.detach().clone() for k, v in sd.items() if '.model.' not in k}


def data(seed, B, QV, A):
    torch.manual_seed(seed)
    img = torch.randn(B, 3, 32, 32)
    qst = torch.randint(QV, (B, 8)); qst[:, 0] = 2
    lab = torch.randint(A, (B,))
    return img, qst, lab


def main(mode):
    stub_tv()
    torch.nn.functional.dropout = lambda input, p=0.5, training=True, inplace=False: input
    dims = dict(embed_size=32, qst_vocab_size=50, ans_vocab_size=40, word_embed_size=16, num_layers=1, hidden_size=32)
    if mode == 'basic':
        sys.path.insert(0, '/root/reference/basic_vqa')
        import config
        config.DEVICE = 'cpu'
        config.ARCH_TYPE = 'darts'
        import models, models_lct
        from pcdarts.architect_lct import ArchitectLct
        torch.manual_seed(10)
        ef = models_lct.VqaModel(**dims)
        torch.manual_seed(12)
        for a in ef.arch_parameters():
            a.data.copy_(torch.randn_like(a) * 0.3)
        torch.manual_seed(20)
        w = models.VqaModel(**dims)
        no_dropout(ef.train()); no_dropout(w.train())
        ef.qst_encoder.max_length = 6
        img, qst, lab = data(30, 2, 50, 40)
        vimg, vqst, vlab = data(31, 2, 50, 40)
        out = {'dims': dims, 'ef_seed': 10, 'w_seed': 20, 'ef_state': small(ef.state_dict()), 'w_state': small(w.state_dict()),
               'arch': [a.detach().clone() for a in ef.arch_parameters()], 'max_length': 6,
               'train': (img, qst, lab), 'valid': (vimg, vqst, vlab)}
        loss = ef._loss(img, qst, lab)
        ps = dict(ef.named_parameters())
        names = ['img_encoder.fc.weight', 'img_encoder.darts.stem.0.weight', 'img_encoder.darts.cells.1._ops.2._ops.6.op.1.weight',
                 'qst_encoder.fc2.weight', 'qst_encoder.lstm.weight_hh_l0', 'fc2.weight']
        grads = torch.autograd.grad(loss, list(ef.arch_parameters()) + [ps[n] for n in names])
        out.update(ef_loss=loss.detach(), ef_arch_grads=[g.clone() for g in grads[:4]],
                   ef_param_grads={n: g.clone() for n, g in zip(names, grads[4:])})
        # the forward above updated the BN running statistics: record the state the architect starts from
        out['ef_state_before_step'] = small(ef.state_dict())
        ef_opt = torch.optim.Adam(ef.parameters(), lr=1e-3)
        w_opt = torch.optim.Adam(w.parameters(), lr=1e-3)
        arch = ArchitectLct(ef, w, ef_opt, w_opt)
        arch.optimizer.zero_grad()
        arch._backward_step_unrolled(img, qst, lab, vimg, vqst, vlab, 1e-3, 1e-3)
        out['lct_arch_grads'] = [a.grad.detach().clone() for a in ef.arch_parameters()]
        arch.optimizer.step()
        out['lct_arch_after'] = [a.detach().clone() for a in ef.arch_parameters()]
        torch.save(out, os.path.join(GOLD, 'arch_lct.pt'))
        print('arch grads abs max', [float(g.abs().max()) for g in out['lct_arch_grads']])
    else:
        sys.path.insert(0, '/root/reference/darts_vqa')
        import config
        config.DEVICE = 'cpu'
        import vqa_model
        from pcdarts.architect_vqa import Architect
        torch.manual_seed(10)
        m = vqa_model.VqaModel(img_encoder_type='darts', **dims)
        torch.manual_seed(12)
        for a in m.arch_parameters():
            a.data.copy_(torch.randn_like(a) * 0.3)
        no_dropout(m.train())
        img, qst, lab = data(30, 2, 50, 40)
        vimg, vqst, vlab = data(31, 2, 50, 40)
        out = {'dims': dims, 'seed': 10, 'state': small(m.state_dict()), 'arch': [a.detach().clone() for a in m.arch_parameters()],
               'train': (img, qst, lab), 'valid': (vimg, vqst, vlab)}
        ans, qo = m(img, qst)
        out.update(ans=ans.detach(), qst_out=qo.detach())
        for qst_only in (False, True):
            m.load_state_dict({**m.state_dict(), **out['state']})
            for a, r in zip(m.arch_parameters(), out['arch']):
                a.data.copy_(r)
                a.grad = None
            args = types.SimpleNamespace(arch_learn_rate=6e-4, arch_wt_decay=1e-3, qst_only=qst_only)
            arch = Architect(m, args)
            out[f'state_before_step_{int(qst_only)}'] = small(m.state_dict())
            arch.optimizer.zero_grad()
            arch._backward_step_unrolled(img, qst, lab, vimg, vqst, vlab, 1e-3, None)
            out[f'arch_grads_{int(qst_only)}'] = [a.grad.detach().clone() for a in m.arch_parameters()]
            arch.optimizer.step()
            out[f'arch_after_{int(qst_only)}'] = [a.detach().clone() for a in m.arch_parameters()]
        torch.save(out, os.path.join(GOLD, 'arch_vqa.pt'))
        print('arch grads abs max', [float(g.abs().max()) for g in out['arch_grads_0']])


if __name__ == '__main__':
    main(sys.argv[1])
