#!/usr/bin/env python3
"""Golden vectors of the PC-DARTS search network, produced by RUNNING the reference's own
basic_vqa/pcdarts/model_search.Network on CPU (imports as-is: no torchvision needed).
    python tests/make_golden_darts.py     # -> tests/golden/darts_net.pt, updates MANIFEST.json
Also checks the CPU oracle (oracle/darts_oracle.py) against the reference while writing (= the pinning run)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/basic_vqa'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def diff(a, b):
    a, b = a.detach().double(), b.detach().double()
    d = (a - b).abs().max().item()
    return d, d / max(b.abs().max().item(), 1e-30)


def main():
    import config
    config.DEVICE = 'cpu'
    from pcdarts.model_search import Network          # reference module
    from oracle import darts_oracle as DO
    torch.manual_seed(10)
    net = Network(16, 512, 4, None)
    torch.manual_seed(11)
    # make the architecture weights non-trivial so that every branch matters
    for a in net.arch_parameters():
        a.data.copy_(torch.randn_like(a) * 0.5)
    img = torch.randn(2, 3, 32, 32)
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    arch = [a.detach().clone() for a in net.arch_parameters()]
    net.eval()
    with torch.no_grad():
        ev = net(img)
    o_ev = DO.network_forward(img, P, arch, False)
    net.train()
    tr = net(img)
    wgt = torch.randn_like(tr)
    loss = (tr * wgt).sum()
    names = ['stem.0.weight', 'stem.1.weight', 'stem.1.bias', 'cells.0.preprocess0.op.1.weight',
             'cells.0._ops.0._ops.4.op.1.weight', 'cells.0._ops.0._ops.4.op.2.weight', 'cells.0._ops.3._ops.7.op.1.weight',
             'cells.1.preprocess0.op.1.weight', 'cells.1._ops.0._ops.3.conv_1.weight', 'cells.1._ops.0._ops.3.conv_2.weight',
             'cells.1._ops.1._ops.5.op.5.weight', 'cells.2.preprocess0.conv_1.weight', 'cells.3._ops.13._ops.6.op.2.weight']
    pd = dict(net.named_parameters())
    grads = torch.autograd.grad(loss, list(net.arch_parameters()) + [pd[n] for n in names])
    after = {k: v.detach().clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}
    # oracle in train mode
    ns = {}
    Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    ag = [a.clone().requires_grad_(True) for a in arch]
    o_tr = DO.network_forward(img, Pg, ag, True, ns)
    o_grads = torch.autograd.grad((o_tr * wgt).sum(), ag + [Pg[n] for n in names])
    man = {'eval_diff': diff(o_ev, ev), 'train_diff': diff(o_tr, tr),
           'grad_rel_diff_max': max(diff(a, b)[1] for a, b in zip(o_grads, grads)),
           'running_stat_diff_max': max(diff(ns[k], after[k])[1] for k in ns)}
    torch.save({'seed': 10, 'C': 16, 'layers': 4, 'state': P, 'arch': arch, 'image': img, 'eval_out': ev,
                'train_out': tr.detach(), 'loss_weight': wgt, 'arch_grads': [g.clone() for g in grads[:4]],
                'param_grads': {n: g.clone() for n, g in zip(names, grads[4:])},
                'running_after': {k: after[k] for k in list(after) if 'num_batches' not in k},
                'genotype': str(net.genotype())}, os.path.join(GOLD, 'darts_net.pt'))
    mf = os.path.join(GOLD, 'MANIFEST.json')
    m = json.load(open(mf))
    m['fixtures']['darts_net'] = man
    json.dump(m, open(mf, 'w'), indent=1)
    print(json.dumps(man, indent=1))


if __name__ == '__main__':
    main()
