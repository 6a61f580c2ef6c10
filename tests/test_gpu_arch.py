"""GPU parity of the DARTS-encoder models and of the architect steps (2-level darts_vqa Architect, 3-level
ArchitectLct) against golden vectors produced by the reference's own classes (tests/make_golden_arch.py)."""
import types

import pytest
import torch

from _util import load, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    return m


def dev(t):
    return tuple(x.to(DEV) for x in t)


def close_arch(got, ref, rtol):
    """architecture gradients: compare relative to the largest entry of the tensor (finite-difference noise)."""
    for a, r in zip(got, ref):
        scale = r.abs().max().item()
        assert (a.cpu() - r).abs().max().item() <= rtol * scale, ((a.cpu() - r).abs().max().item(), scale)


def test_darts_vqa_model_and_architect(hip_lib):
    from lct_vqa_amd.darts_vqa import config, vqa_model
    from lct_vqa_amd.basic_vqa import config as bconfig
    config.DEVICE = bconfig.DEVICE = torch.device(DEV)
    from lct_vqa_amd.darts_vqa.pcdarts.architect_vqa import Architect
    from lct_vqa_amd import ops
    fx = load('arch_vqa.pt')
    torch.manual_seed(fx['seed'])
    m = vqa_model.VqaModel(img_encoder_type='darts', **fx['dims'])
    assert set(m.state_dict()) == set(fx['state'])
    for k, v in fx['state'].items():                       # identical init for the same seed
        assert torch.equal(m.state_dict()[k].cpu(), v), k
    m = no_dropout(m.to(DEV).train())
    for a, r in zip(m.arch_parameters(), fx['arch']):
        a.data.copy_(r)
    img, qst, lab = dev(fx['train'])
    vimg, vqst, vlab = dev(fx['valid'])
    ans, qo = m(img, qst)
    assert rel_err(ans, fx['ans']) < 1e-3 and rel_err(qo, fx['qst_out']) < 1e-3
    with ops.dropout_disabled():
        _arch_vqa_steps(m, fx, Architect, img, qst, lab, vimg, vqst, vlab)


def _arch_vqa_steps(m, fx, Architect, img, qst, lab, vimg, vqst, vlab):
    for qst_only in (0, 1):
        m.load_state_dict({**m.state_dict(), **{k: v.to(DEV) for k, v in fx[f'state_before_step_{qst_only}'].items()}})
        for a, r in zip(m.arch_parameters(), fx['arch']):
            a.data.copy_(r)
            a.grad = None
        args = types.SimpleNamespace(arch_learn_rate=6e-4, arch_wt_decay=1e-3, qst_only=bool(qst_only))
        arch = Architect(m, args)
        arch.optimizer.zero_grad()
        arch.compute_grads(img, qst, lab, vimg, vqst, vlab, 1e-3)                     # asserts the None-count (0 / 6)
        close_arch([a.grad for a in m.arch_parameters()], fx[f'arch_grads_{qst_only}'], 2e-2)
        arch.optimizer.step()
        for a, r in zip(m.arch_parameters(), fx[f'arch_after_{qst_only}']):
            assert (a.detach().cpu() - r).abs().max().item() < 1.5e-4          # Adam first step = +-6e-4


def test_ef_darts_model_and_lct_architect(hip_lib):
    from lct_vqa_amd.basic_vqa import config, models, models_lct
    config.DEVICE = torch.device(DEV)
    config.ARCH_TYPE = 'darts'
    try:
        from lct_vqa_amd.basic_vqa.pcdarts.architect_lct import ArchitectLct
        from lct_vqa_amd.optim import FusedAdam
        fx = load('arch_lct.pt')
        torch.manual_seed(fx['ef_seed'])
        ef = models_lct.VqaModel(**fx['dims'])
        assert set(ef.state_dict()) == set(fx['ef_state']) and len(ef.state_dict()) > 2100
        for k, v in fx['ef_state'].items():
            assert torch.equal(ef.state_dict()[k].cpu(), v), k
        torch.manual_seed(fx['w_seed'])
        w = models.VqaModel(**fx['dims'])
        for k, v in fx['w_state'].items():
            assert torch.equal(w.state_dict()[k].cpu(), v), k
        ef, w = no_dropout(ef.to(DEV).train()), no_dropout(w.to(DEV).train())
        ef.qst_encoder.max_length = fx['max_length']
        for a, r in zip(ef.arch_parameters(), fx['arch']):
            a.data.copy_(r)
        img, qst, lab = dev(fx['train'])
        vimg, vqst, vlab = dev(fx['valid'])
        loss = ef._loss(img, qst, lab)
        assert rel_err(loss, fx['ef_loss']) < 1e-4
        ps = dict(ef.named_parameters())
        names = list(fx['ef_param_grads'])
        grads = torch.autograd.grad(loss, list(ef.arch_parameters()) + [ps[n] for n in names])
        close_arch(grads[:4], fx['ef_arch_grads'], 2e-3)
        for n, g in zip(names, grads[4:]):
            assert rel_err(g, fx['ef_param_grads'][n]) < 2e-3, n
        ef.load_state_dict({**ef.state_dict(), **{k: v.to(DEV) for k, v in fx['ef_state_before_step'].items()}})
        from lct_vqa_amd import ops
        arch = ArchitectLct(ef, w, FusedAdam(ef.parameters(), lr=1e-3), FusedAdam(w.parameters(), lr=1e-3))
        arch.optimizer.zero_grad()
        with ops.dropout_disabled():
            arch.compute_grads(img, qst, lab, vimg, vqst, vlab, 1e-3, 1e-3)           # asserts None-counts 0 / 36 / 2
        close_arch([a.grad for a in ef.arch_parameters()], fx['lct_arch_grads'], 0.25)
        arch.optimizer.step()
        for a, r in zip(ef.arch_parameters(), fx['lct_arch_after']):
            assert (a.detach().cpu() - r).abs().max().item() < 6.5e-4
    finally:
        config.ARCH_TYPE = 'fixed'
