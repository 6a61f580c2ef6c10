#!/usr/bin/env python3
"""Per-stage deviation of the HIP ArchitectLct step from the reference's recorded intermediates
(tests/golden/arch_lct.pt, made by tests/make_golden_arch.py basic).  Diagnostic: prints, asserts nothing."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _util import load  # noqa: E402


def relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    from lct_vqa_amd import ops
    from lct_vqa_amd._lib import lib
    from lct_vqa_amd.basic_vqa import config, models, models_lct
    from lct_vqa_amd.basic_vqa.pcdarts.architect_lct import ArchitectLct
    from lct_vqa_amd.optim import FusedAdam
    lib.load()
    if len(sys.argv) > 1:
        ops.set_conv_algo(sys.argv[1])
    DEV = 'cuda'
    config.DEVICE = torch.device(DEV)
    config.ARCH_TYPE = 'darts'
    fx = load('arch_lct.pt')
    torch.manual_seed(fx['ef_seed'])
    ef = models_lct.VqaModel(**fx['dims'])
    torch.manual_seed(fx['w_seed'])
    w = models.VqaModel(**fx['dims'])
    for m in (ef, w):
        m.to(DEV).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.eval()
    ef.qst_encoder.max_length = fx['max_length']
    for a, r in zip(ef.arch_parameters(), fx['arch']):
        a.data.copy_(r)
    ef.load_state_dict({**ef.state_dict(), **{k: v.to(DEV) for k, v in fx['ef_state_before_step'].items()}})
    d = lambda t: tuple(x.to(DEV) for x in t)
    arch = ArchitectLct(ef, w, FusedAdam(ef.parameters(), lr=1e-3), FusedAdam(w.parameters(), lr=1e-3))
    arch.trace = {}
    with ops.dropout_disabled():
        arch.compute_grads(*d(fx['train']), *d(fx['valid']), 1e-3, 1e-3)
    tr, ref, f64 = arch.trace, fx['lct'], fx['lct_f64']
    print('pseudo_qst equal', torch.equal(tr['pseudo_qst'].cpu(), ref['pseudo_qst']))
    print('pseudo_ans', relmax(tr['pseudo_ans'], ref['pseudo_ans']))
    print('val_loss', float(tr['val_loss']), float(ref['val_loss']), float(f64['val_loss']))
    efa = dict(tr['ef_ahead'].named_parameters())
    print('ef_ahead', max(relmax(efa[n], t) for n, t in ref['ef_ahead'].items()))
    wa = dict(tr['w_ahead'].named_parameters())
    print('w_ahead', max(relmax(wa[n], t) for n, t in ref['w_ahead'].items()))
    gw = {n: g for (n, _), g in zip(w.named_parameters(), tr['grad_wprime'])}
    for n, t in ref['grad_wprime'].items():
        print('  grad_wprime', n, relmax(gw[n], t), 'ref32-vs-64', relmax(t, f64['grad_wprime'][n]))
    kp = {n: g for (n, _), g in zip(ef.named_parameters(), tr['kappa'])}
    worst = sorted(((relmax(kp[n], t), n) for n, t in ref['kappa'].items() if t.abs().max() > 0), reverse=True)
    print('kappa worst', worst[:5])
    num = sum(float((kp[n].cpu().double() - t.double()).pow(2).sum()) for n, t in ref['kappa'].items())
    den = sum(float(t.double().pow(2).sum()) for t in ref['kappa'].values())
    print('kappa global rel L2', (num / den) ** 0.5, ' ref32-vs-64', f64['kappa_f32_rel_l2'], 'worst tensor', f64['kappa_f32_worst_tensor'])
    for i, (g, r, r64, r3) in enumerate(zip(tr['gamma'], ref['gamma'], f64['gamma'], fx['lct_gamma_r3'])):
        print('  gamma', i, 'gpu-vs-ref32', relmax(g, r), 'gpu-vs-ref64', relmax(g, r64), 'ref32-vs-64', relmax(r, r64), 'r3-vs-r1', relmax(r3, r))




def two_level():
    """Same for the 2-level basic_vqa Architect (tests/golden/arch_two_level.pt)."""
    from lct_vqa_amd import ops, bilevel
    from lct_vqa_amd._lib import lib
    from lct_vqa_amd.basic_vqa import config, models_lct
    from lct_vqa_amd.basic_vqa.pcdarts.architect import Architect
    lib.load()
    DEV = 'cuda'
    config.DEVICE = torch.device(DEV)
    config.ARCH_TYPE = 'darts'
    fx = load('arch_two_level.pt')
    torch.manual_seed(fx['ef_seed'])
    ef = models_lct.VqaModel(**fx['dims']).to(DEV).train()
    for mod in ef.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    for a, r in zip(ef.arch_parameters(), fx['arch']):
        a.data.copy_(r)
    d = lambda t: tuple(x.to(DEV) for x in t)
    img, qst, lab = d(fx['train'])
    with ops.dropout_disabled():
        # repeatability and accuracy of the architecture gradient itself
        g1 = bilevel.grads(ef._loss(img, qst, lab), ef.arch_parameters())
        g2 = bilevel.grads(ef._loss(img, qst, lab), ef.arch_parameters())
        print('alpha grad run-to-run', [relmax(a, b) for a, b in zip(g1, g2)])
        lf = load('arch_lct.pt')
        print('alpha grad vs reference (same model, arch_lct.pt train batch)', [relmax(a, b) for a, b in zip(g1, lf['ef_arch_grads'])])
        arch = Architect(ef)
        arch.trace = {}
        arch.compute_grads(img, qst, lab, *d(fx['valid']), 1e-3, True)
    ref = fx['unrolled']
    tr = arch.trace
    print('dalpha_direct', [relmax(a, b) for a, b in zip(tr['dalpha_direct'], ref['dalpha_direct'])])
    v = {n: g for (n, _), g in zip(ef.named_parameters(), tr['v'])}
    for n, t in ref['v'].items():
        print('  v', n, relmax(v[n], t))
    print('implicit gpu-vs-ref32', [relmax(a, b) for a, b in zip(tr['implicit'], ref['implicit'])])
    print('implicit ref32-vs-ref64', [relmax(a, b) for a, b in zip(ref['implicit'], ref['implicit_f64'])])
    print('arch_grads', [relmax(a.grad, b) for a, b in zip(ef.arch_parameters(), ref['arch_grads'])])


if __name__ == '__main__':
    two_level() if sys.argv[1:] == ['two'] else main()
