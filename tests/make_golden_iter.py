#!/usr/bin/env python3
"""Golden vectors of whole TRAIN ITERATIONS of the LCT driver, produced by the reference's own modules on CPU.

    python tests/make_golden_iter.py        # -> tests/golden/train_iter.pt

What one iteration of reference basic_vqa/experiment.py:216-257 does to the two models, replayed with the reference's
model_factory / architect_factory (imported from /root/reference, never copied):
  stage 3 (ARCH_TYPE 'darts' only, first batch)   architect.step(train batch, valid batch, ef_lr, w_lr)   (:199-212)
  stage 1   EF forward, CE(answers) + CE(next tokens), backward, clip_grad_norm_ over ALL parameters, Adam   (:216-227)
  stage 2   EF.generate(image) with grad enabled, softmax of its answers, W on the real and on the generated
            questions, CE + W_LAMBDA * softXEnt, backward, clip, Adam                                        (:242-257)
The driver itself (experiment.py) needs nltk, h5py and the VQA data set and cannot be imported here; the sequence above is
this script's replay of its train loop body, every object in it is the reference's.  nn.Dropout is neutralised as in
tests/make_golden_arch.py (the CPU Bernoulli stream cannot be replayed on the GPU).  Two iterations are recorded for the
fixed (frozen VGG19) examiner so that the second one exercises Adam's moments; one iteration incl. stage 3 for 'darts'.
torchvision.models.vgg19 = oracle/tv_vgg19.py (see tests/make_golden.py)."""
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from make_golden_arch import stub_tv, small  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def batch(seed, B, T, QV, A, S):
    torch.manual_seed(seed)
    img = torch.randn(B, 3, S, S)
    qst = torch.randint(QV, (B, T)); qst[:, 0] = 2
    return img, qst, torch.randint(A, (B,))


PICK = ('img_encoder.darts.stem.0.weight', 'img_encoder.darts.cells.0.preprocess1.op.1.weight',
        'img_encoder.darts.cells.1._ops.2._ops.6.op.1.weight', 'img_encoder.darts.cells.3._ops.13._ops.4.op.5.weight',
        'img_encoder.darts.cells.2._ops.0._ops.3.conv_1.weight', 'img_encoder.darts.cells.3._ops.13._ops.4.op.3.running_var')


def changed(model, before):
    """Tensors the iteration changed (VGG19 excluded; of the 2100 tensors of a DARTS encoder a few named ones)."""
    return {k: v.detach().clone() for k, v in model.state_dict().items()
            if '.model.' not in k and not torch.equal(v, before[k]) and ('.darts.' not in k or k in PICK)}


def main():
    stub_tv()
    torch.nn.functional.dropout = lambda input, p=0.5, training=True, inplace=False: input
    sys.path.insert(0, '/root/reference/basic_vqa')
    import config
    config.DEVICE = 'cpu'
    config.IMG_EMBED_SIZE, config.WORD_EMBED_SIZE, config.LSTM_HIDDEN_SIZE = 32, 16, 32
    config.GRAD_CLIP = 0.5                # below the gradient norms of these batches (0.7 ... 1.5): the clip is live
    import model_factory as MF
    import architect_factory as AF
    from models import softXEnt
    QV, A, B, T, S = 50, 40, 4, 14, 32
    dataset = types.SimpleNamespace(qst_vocab=types.SimpleNamespace(vocab_size=QV), ans_vocab=types.SimpleNamespace(vocab_size=A))
    criterion = nn.CrossEntropyLoss()
    out = {'grad_clip': config.GRAD_CLIP, 'sizes': dict(QV=QV, A=A, B=B, T=T, S=S, embed=32, word_embed=16, hidden=32), 'seed': config.SEED}
    for arch_type, n_iter in (('fixed', 2), ('darts', 1)):
        config.ARCH_TYPE, config.SKIP_STAGE2, config.SKIP_STAGE3 = arch_type, False, False
        torch.manual_seed(config.SEED)
        ef_model = MF.get_ef_model(dataset); ef_optimizer = MF.get_ef_optimizer(ef_model); ef_scheduler = MF.get_ef_scheduler(ef_optimizer)
        w_model = MF.get_w_model(dataset); w_optimizer = MF.get_w_optimizer(w_model)
        architect = AF.get_architect(ef_model, w_model, ef_optimizer, w_optimizer)
        if arch_type == 'darts':
            torch.manual_seed(12)
            for a in ef_model.arch_parameters():
                a.data.copy_(torch.randn_like(a) * 0.3)
        ef_model.train(); w_model.train()
        rec = {'w_state': small(w_model.state_dict()), 'iters': []}
        if arch_type == 'darts':      # the examiner is arch_lct.pt's (seed 10, same sizes): its initial state is pinned there
            rec['arch'] = [a.detach().clone() for a in ef_model.arch_parameters()]
        else:
            rec['ef_state'] = small(ef_model.state_dict())
        for it in range(n_iter):
            image, question, label = batch(100 + it, B, T, QV, A, S)
            ef_before = {k: v.clone() for k, v in ef_model.state_dict().items()}
            w_before = {k: v.clone() for k, v in w_model.state_dict().items()}
            r = {'batch': (image, question, label)}
            ef_lr = w_lr = ef_scheduler.get_last_lr()[0]                         # experiment.py:188-189
            if architect is not None and it == 0:
                val = batch(200 + it, B, T, QV, A, S)
                r['valid'] = val
                architect.step(image, question, label, *val, ef_lr, w_lr)
                r['arch_after'] = [a.detach().clone() for a in ef_model.arch_parameters()]
            # stage 1
            ef_optimizer.zero_grad()
            ans_out, qst_out = ef_model(image, question)
            loss = criterion(ans_out, label) + criterion(qst_out[:, :-1].flatten(end_dim=1), question[:, 1:].flatten())
            loss.backward()
            r['ef_grad_norm'] = nn.utils.clip_grad_norm_(ef_model.parameters(), config.GRAD_CLIP).detach().clone()
            ef_optimizer.step()
            r.update(ef_loss=loss.detach().clone(), ef_pred=ans_out.argmax(1))
            # stage 2
            pseudo_qst, pseudo_ans = ef_model.generate(image)
            pseudo_ans = F.softmax(pseudo_ans, dim=1)
            w_optimizer.zero_grad()
            w_out_1 = w_model(image, question)
            w_out_2 = w_model(image, pseudo_qst)
            loss = criterion(w_out_1, label) + config.W_LAMBDA * softXEnt(w_out_2, pseudo_ans)
            loss.backward()
            r['w_grad_norm'] = nn.utils.clip_grad_norm_(w_model.parameters(), config.GRAD_CLIP).detach().clone()
            w_optimizer.step()
            r.update(pseudo_qst=pseudo_qst.clone(), pseudo_ans=pseudo_ans.detach().clone(), w_loss=loss.detach().clone(),
                     w_pred_1=w_out_1.argmax(1), w_pred_2=w_out_2.argmax(1),
                     ef_after=changed(ef_model, ef_before), w_after=changed(w_model, w_before))
            rec['iters'].append(r)
            print(arch_type, it, 'ef_loss', float(r['ef_loss']), 'w_loss', float(r['w_loss']), 'norms', float(r['ef_grad_norm']), float(r['w_grad_norm']),
                  'changed', len(r['ef_after']), len(r['w_after']))
        out[arch_type] = rec
    torch.save(out, os.path.join(GOLD, 'train_iter.pt'))


if __name__ == '__main__':
    main()
