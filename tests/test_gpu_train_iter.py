"""Whole train iterations of the LCT driver (reference basic_vqa/experiment.py:199-257: stage 3 architect step, stage 1
examiner update, stage 2 generate + student update) replayed through THIS package's model_factory / architect_factory,
with torch's own ``clip_grad_norm_`` over all parameters and the factory's optimizer exactly as the reference's driver
would call them, against what the reference's modules produced for the same batches (tests/make_golden_iter.py)."""
import types

import pytest
import torch
import torch.nn as nn

from _util import load, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _after_close(model, ref_after, lr, name):
    """Parameters after clip + Adam.  Adam's update is -lr * m / (sqrt(v) + eps): where a gradient entry is rounding
    noise around zero its SIGN decides a full +-lr step, so single entries may differ by up to 2 * lr between two correct
    fp32 implementations; everything else must agree to a small fraction of one step."""
    sd = model.state_dict()
    for k, r in ref_after.items():
        g = sd[k].detach().cpu()
        if not r.is_floating_point():
            assert torch.equal(g, r), (name, k)
            continue
        d = (g - r).abs()
        outliers = (d > 0.02 * lr).float().mean().item()
        assert outliers <= 2e-3, (name, k, outliers)
        assert d.max().item() <= 2.0 * lr * 1.001 + 1e-6 * r.abs().max().item(), (name, k, d.max().item())
        assert d.mean().item() <= 2e-3 * lr, (name, k, d.mean().item())


@pytest.mark.parametrize('arch_type', ['fixed', 'darts'])
def test_three_stage_iteration_replay(hip_lib, arch_type):
    from lct_vqa_amd import ops
    from lct_vqa_amd.basic_vqa import config, model_factory as MF, architect_factory as AF
    from lct_vqa_amd.basic_vqa.models import softXEnt
    fx = load('train_iter.pt')
    sz, rec = fx['sizes'], fx[arch_type]
    saved = {k: getattr(config, k) for k in ('DEVICE', 'ARCH_TYPE', 'SKIP_STAGE2', 'SKIP_STAGE3', 'IMG_EMBED_SIZE',
                                              'WORD_EMBED_SIZE', 'LSTM_HIDDEN_SIZE', 'GRAD_CLIP')}
    try:
        config.DEVICE = torch.device(DEV)
        config.ARCH_TYPE, config.SKIP_STAGE2, config.SKIP_STAGE3 = arch_type, False, False
        config.IMG_EMBED_SIZE, config.WORD_EMBED_SIZE, config.LSTM_HIDDEN_SIZE = sz['embed'], sz['word_embed'], sz['hidden']
        config.GRAD_CLIP = fx['grad_clip']
        dataset = types.SimpleNamespace(qst_vocab=types.SimpleNamespace(vocab_size=sz['QV']),
                                        ans_vocab=types.SimpleNamespace(vocab_size=sz['A']))
        torch.manual_seed(fx['seed'])                       # experiment.py:35-38, then the factories in the driver's order
        ef_model = MF.get_ef_model(dataset)
        w_model = MF.get_w_model(dataset)
        ef_state = rec['ef_state'] if arch_type == 'fixed' else load('arch_lct.pt')['ef_state']
        for k, v in ef_state.items():                       # same seed, same construction order: identical parameters
            assert torch.equal(ef_model.state_dict()[k], v), k
        for k, v in rec['w_state'].items():
            assert torch.equal(w_model.state_dict()[k], v), k
        ef_model, w_model = ef_model.to(DEV).train(), w_model.to(DEV).train()       # experiment.py:120-123
        ef_optimizer, w_optimizer = MF.get_ef_optimizer(ef_model), MF.get_w_optimizer(w_model)
        ef_scheduler = MF.get_ef_scheduler(ef_optimizer)
        architect = AF.get_architect(ef_model, w_model, ef_optimizer, w_optimizer)
        assert (architect is None) == (arch_type == 'fixed')
        if arch_type == 'darts':
            for a, r in zip(ef_model.arch_parameters(), rec['arch']):
                a.data.copy_(r)
        criterion = nn.CrossEntropyLoss()
        with ops.dropout_disabled():
            for it, r in enumerate(rec['iters']):
                image, question, label = (t.to(DEV) for t in r['batch'])
                ef_lr = w_lr = ef_scheduler.get_last_lr()[0]
                if architect is not None and it == 0:
                    before = [a.detach().clone() for a in ef_model.arch_parameters()]
                    architect.step(image, question, label, *(t.to(DEV) for t in r['valid']), ef_lr, w_lr)
                    for b, a, rb, ra in zip(before, ef_model.arch_parameters(), rec['arch'], r['arch_after']):
                        assert ((a.detach().cpu() - b.cpu()) - (ra - rb)).abs().max().item() <= 0.02 * 6e-4
                # stage 1
                ef_optimizer.zero_grad()
                ans_out, qst_out = ef_model(image, question)
                loss = criterion(ans_out, label) + criterion(qst_out[:, :-1].flatten(end_dim=1), question[:, 1:].flatten())
                loss.backward()
                norm = nn.utils.clip_grad_norm_(ef_model.parameters(), config.GRAD_CLIP)
                ef_optimizer.step()
                assert abs(loss.item() - r['ef_loss'].item()) <= 1e-5 * r['ef_loss'].item()
                assert abs(norm.item() - r['ef_grad_norm'].item()) <= 1e-3 * r['ef_grad_norm'].item()
                assert torch.equal(ans_out.argmax(1).cpu(), r['ef_pred'])
                # stage 2
                pseudo_qst, pseudo_ans = ef_model.generate(image)
                pseudo_ans = torch.softmax(pseudo_ans, dim=1)
                assert torch.equal(pseudo_qst.cpu(), r['pseudo_qst'])                  # generated tests: bit-exact ids
                assert rel_err(pseudo_ans, r['pseudo_ans']) < 1e-3
                w_optimizer.zero_grad()
                w_out_1 = w_model(image, question)
                w_out_2 = w_model(image, pseudo_qst)
                loss = criterion(w_out_1, label) + config.W_LAMBDA * softXEnt(w_out_2, pseudo_ans)
                loss.backward()
                norm = nn.utils.clip_grad_norm_(w_model.parameters(), config.GRAD_CLIP)
                w_optimizer.step()
                assert abs(loss.item() - r['w_loss'].item()) <= 1e-5 * r['w_loss'].item()
                assert abs(norm.item() - r['w_grad_norm'].item()) <= 1e-3 * r['w_grad_norm'].item()
                assert torch.equal(w_out_1.argmax(1).cpu(), r['w_pred_1']) and torch.equal(w_out_2.argmax(1).cpu(), r['w_pred_2'])
                _after_close(ef_model, r['ef_after'], ef_lr, f'ef[{it}]')
                _after_close(w_model, r['w_after'], w_lr, f'w[{it}]')
    finally:
        for k, v in saved.items():
            setattr(config, k, v)
