#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE's own
modules (imported from /root/reference, never copied) on CPU in the build container.

    python tests/make_golden.py            # writes tests/golden/*.pt + MANIFEST.json

The reference imports ``torchvision.models`` (basic_vqa/models.py:3), which is not
installed here.  torchvision is a third-party, un-vendored dependency; its VGG19
*architecture* is restated in oracle/tv_vgg19.py and registered as
``torchvision.models`` before the import.  ``pretrained=True`` cannot be honoured
offline: weights are the seeded random init of that restatement, so VGG19 itself is
"parity unpinned by the reference"; every reference-owned line on the path
(models.py, models_lct.py, pcdarts/*) runs unmodified.

While writing each fixture the script also checks the CPU oracle
(oracle/vqa_oracle.py) against the reference output and records the max abs / rel
difference in MANIFEST.json -- this is the run that pins the oracle.

Fixtures hold DATA only (inputs, small parameter tensors, expected outputs,
checksums); large parameter tensors (VGG19, 575 MB) are regenerated from the
recorded seed and verified by checksum.
"""
import json
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/basic_vqa'
sys.path.insert(0, ROOT)
from oracle import tv_vgg19, vqa_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvm.vgg19 = tv_vgg19.vgg19
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    sys.path.insert(0, REF)
    import config  # reference config
    config.DEVICE = 'cpu'
    config.ARCH_TYPE = 'fixed'
    import models, models_lct  # noqa: E401  (reference modules)
    return config, models, models_lct


def checksum(t):
    t = t.detach().double()
    return [float(t.sum()), float(t.abs().sum())]


def diff(a, b):
    a, b = a.detach().double(), b.detach().double()
    d = (a - b).abs().max().item()
    r = d / max(b.abs().max().item(), 1e-30)
    return d, r


def small_state(model):
    """state_dict split: small tensors kept verbatim, VGG tensors as checksums."""
    keep, sums = {}, {}
    for k, v in model.state_dict().items():
        if '.model.' in k:
            sums[k] = checksum(v)
        else:
            keep[k] = v.clone()
    return keep, sums


def main():
    torch.set_num_threads(8)
    config, models, models_lct = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    manifest = {'torch': torch.__version__, 'fixtures': {}}

    # ------------------------------------------------------------------ losses
    torch.manual_seed(1)
    pred = torch.randn(6, 17)
    tgt = F.softmax(torch.randn(6, 17) / 0.1, dim=1)
    lab = torch.randint(17, (6,))
    ref_soft = models.softXEnt(pred, tgt)
    ref_ce = torch.nn.CrossEntropyLoss()(pred, lab)
    torch.save({'pred': pred, 'target': tgt, 'label': lab, 'softxent': ref_soft, 'ce': ref_ce},
               os.path.join(GOLD, 'losses.pt'))
    manifest['fixtures']['losses'] = {
        'softxent_diff': diff(O.soft_xent(pred, tgt), ref_soft),
        'ce_diff': diff(O.cross_entropy(pred, lab), ref_ce)}

    # --------------------------------------------------------------- W, small
    dims = dict(embed_size=32, qst_vocab_size=50, ans_vocab_size=40,
                word_embed_size=16, num_layers=1, hidden_size=32)
    seed = 10
    torch.manual_seed(seed)
    w = models.VqaModel(**dims)
    B, T, S = 4, 14, 32
    torch.manual_seed(seed + 1)
    img = torch.randn(B, 3, S, S)
    qst = torch.randint(dims['qst_vocab_size'], (B, T))
    qst[:, 0] = 2
    lab = torch.randint(dims['ans_vocab_size'], (B,))
    P = {k: v.detach() for k, v in w.state_dict().items()}
    keep, sums = small_state(w)
    w.eval()
    with torch.no_grad():
        ev = w(img, qst)
    o_ev = O.w_model_forward(img, qst, P)
    # train mode: identical masks via identical generator consumption
    w.train()
    torch.manual_seed(77)
    tr = w(img, qst)
    loss = w.criterion(tr, lab)
    grads = torch.autograd.grad(loss, list(w.parameters()), allow_unused=True)
    n_none = sum(g is None for g in grads)
    gd = {k: g.clone() for (k, _), g in zip(w.named_parameters(), grads) if g is not None}
    torch.manual_seed(77)
    masks = O.draw_masks(O.w_mask_shapes(B, dims['ans_vocab_size'], dims['embed_size']))
    Pg = {k: (v.clone().requires_grad_(True) if k in gd else v) for k, v in P.items()}
    o_tr = O.w_model_forward(img, qst, Pg, masks)
    o_loss = O.cross_entropy(o_tr, lab)
    o_grads = torch.autograd.grad(o_loss, [Pg[k] for k in gd])
    gdiff = max(diff(a, gd[k])[1] for a, k in zip(o_grads, gd))
    # soft loss (models.py:117-125), eval mode to avoid mask bookkeeping
    w.eval()
    pq = torch.randint(dims['qst_vocab_size'], (B, 30))
    pl = F.softmax(torch.randn(B, dims['ans_vocab_size']) / config.TEMPERATURE, dim=1)
    with torch.no_grad():
        sl = w._soft_loss(img, qst, lab, pq, pl)
    o_sl = O.w_soft_loss(img, qst, lab, pq, pl, P, config.W_LAMBDA)
    torch.save({'dims': dims, 'seed': seed, 'image': img, 'question': qst, 'label': lab,
                'state': keep, 'vgg_checksums': sums, 'eval_logits': ev,
                'train_masks': [m.to(torch.uint8) for m in masks], 'train_logits': tr.detach(),
                'train_loss': loss.detach(), 'grads': gd, 'num_none_grads': n_none,
                'pseudo_qst': pq, 'pseudo_label': pl, 'soft_loss': sl, 'w_lambda': config.W_LAMBDA},
               os.path.join(GOLD, 'w_small.pt'))
    manifest['fixtures']['w_small'] = {
        'eval_diff': diff(o_ev, ev), 'train_diff': diff(o_tr, tr), 'loss_diff': diff(o_loss, loss),
        'grad_rel_diff_max': gdiff, 'num_none_grads': n_none, 'soft_loss_diff': diff(o_sl, sl)}
    assert n_none == 36

    # -------------------------------------------------------------- EF, small
    dims_ef = dict(dims)
    for pretrained in (True, False):
        torch.manual_seed(seed)
        ef = models_lct.VqaModel(pretrained=pretrained, **dims_ef)
        P = {k: v.detach() for k, v in ef.state_dict().items()}
        keep, sums = small_state(ef)
        ef.eval()
        with torch.no_grad():
            ans, qo = ef(img, qst)
            gq, ga = ef.generate(img)
        o_ans, o_qo = O.ef_model_forward(img, qst, P, None, pretrained)
        o_gq, o_ga = O.ef_generate(img, P, None, pretrained, ef.qst_encoder.max_length)
        loss = ef._loss(img, qst, lab)
        grads = torch.autograd.grad(loss, list(ef.parameters()), allow_unused=True)
        n_none = sum(g is None for g in grads)
        gd, gsum = {}, {}
        for (k, _), g in zip(ef.named_parameters(), grads):
            if g is None:
                continue
            if '.model.' in k:
                gsum[k] = checksum(g)
            else:
                gd[k] = g.clone()
        Pg = {k: (v.clone().requires_grad_(True) if k in gd else v) for k, v in P.items()}
        o_loss = O.ef_loss(img, qst, lab, Pg, None, pretrained)
        o_grads = torch.autograd.grad(o_loss, [Pg[k] for k in gd])
        gdiff = max(diff(a, gd[k])[1] for a, k in zip(o_grads, gd))
        name = 'ef_small_frozen' if pretrained else 'ef_small_trainvgg'
        torch.save({'dims': dims_ef, 'seed': seed, 'pretrained': pretrained, 'image': img,
                    'question': qst, 'label': lab, 'state': keep, 'vgg_checksums': sums,
                    'eval_ans': ans, 'eval_qst_out': qo, 'gen_qst': gq, 'gen_ans': ga,
                    'loss': loss.detach(), 'grads': gd, 'vgg_grad_checksums': gsum,
                    'num_none_grads': n_none}, os.path.join(GOLD, name + '.pt'))
        manifest['fixtures'][name] = {
            'ans_diff': diff(o_ans, ans), 'qst_out_diff': diff(o_qo, qo),
            'gen_tokens_equal': bool((o_gq == gq).all()), 'gen_ans_diff': diff(o_ga, ga),
            'loss_diff': diff(o_loss, loss), 'grad_rel_diff_max': gdiff, 'num_none_grads': n_none}
        assert (o_gq == gq).all()

    # ------------------------------------------- W, full size (BASELINE config 1)
    full = dict(embed_size=512, qst_vocab_size=17858, ans_vocab_size=3129,
                word_embed_size=300, num_layers=1, hidden_size=512)
    torch.manual_seed(seed)
    w = models.VqaModel(**full).eval()
    B = 4
    torch.manual_seed(seed)                      # seed 10 for data as well
    img = torch.randn(B, 3, 224, 224)
    qst = torch.randint(full['qst_vocab_size'], (B, 14))
    qst[:, 0] = 2
    lab = torch.randint(full['ans_vocab_size'], (B,))
    with torch.no_grad():
        lg = w(img, qst)
    P = {k: v.detach() for k, v in w.state_dict().items()}
    o_lg = O.w_model_forward(img, qst, P)
    top2 = lg.topk(2, dim=1).values
    torch.save({'dims': full, 'seed': seed, 'data_seed': seed, 'B': B,
                'param_checksums': {k: checksum(v) for k, v in P.items()},
                'image_checksum': checksum(img), 'question': qst, 'label': lab,
                'eval_logits': lg, 'argmax': lg.argmax(1), 'top2_margin': top2[:, 0] - top2[:, 1]},
               os.path.join(GOLD, 'w_full_b4.pt'))
    manifest['fixtures']['w_full_b4'] = {
        'eval_diff': diff(o_lg, lg), 'argmax_equal': bool((o_lg.argmax(1) == lg.argmax(1)).all()),
        'top2_margin_min': float((top2[:, 0] - top2[:, 1]).min()),
        'logit_absmax': float(lg.abs().max())}

    with open(os.path.join(GOLD, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == '__main__':
    main()
