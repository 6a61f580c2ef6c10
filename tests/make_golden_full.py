#!/usr/bin/env python3
"""Full-size golden of the LCT examiner model (BASELINE configs[3]'s EF model at B=4, 224x224, T=14, QV=17 858, A=3129),
made by RUNNING THE REFERENCE's own ``models_lct.VqaModel`` (imported from /root/reference, never copied) on CPU:

    python tests/make_golden_full.py          # writes tests/golden/ef_full_b4.pt, updates MANIFEST.json

The analogue of ``w_full_b4.pt`` (tests/make_golden.py) for reference basic_vqa/models_lct.py:200-257: eval-mode answer
logits, per-token logits ``qst_out`` (argmax per position, top-2 margins, checksums, a strided sample of entries), the ids the
greedy ``generate`` emits and its answer logits, and ``_loss`` with the gradient checksums of every trainable tensor.
Weights are the seeded default init (575 MB of VGG19 cannot be committed): the GPU test re-creates them with the same seed
through this package's module shells and verifies every tensor by checksum.  The oracle (oracle/vqa_oracle.py) is checked
against the same run; differences go to MANIFEST.json.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from make_golden import import_reference, checksum, diff, GOLD  # noqa: E402
from oracle import vqa_oracle as O  # noqa: E402


def main():
    torch.set_num_threads(8)
    config, models, models_lct = import_reference()
    full = dict(embed_size=512, qst_vocab_size=17858, ans_vocab_size=3129, word_embed_size=300, num_layers=1, hidden_size=512)
    seed, B, T = 10, 4, 14
    torch.manual_seed(seed)
    ef = models_lct.VqaModel(pretrained=True, **full).eval()          # pretrained=True: frozen trunk (models_lct.py:56-61)
    P = {k: v.detach() for k, v in ef.state_dict().items()}
    # Argmax parity needs decisive choices ("fixtures chosen with margin > 1e-2 |logit|"): with random weights the
    # 30 greedy steps over 17 858 words have top-2 gaps down to 1e-5 of the logits for some images, and one flipped token
    # changes every later one.  Scan a few data seeds with the oracle and keep the most decisive batch; its margins are
    # stored so the tests can state what they rely on.
    best = None
    with torch.no_grad():
        for ds in range(seed + 5, seed + 25):
            torch.manual_seed(ds)
            im = torch.randn(B, 3, 224, 224)
            f = O.img_encoder(im, P, None, frozen=True)
            mg = []
            O.lct_generate_tokens(f, P, ef.qst_encoder.max_length, margins=mg)
            m = float(torch.stack(mg).min())
            print('data seed', ds, 'min greedy margin', m, flush=True)
            if best is None or m > best[0]:
                best = (m, ds)
    data_seed = best[1]
    torch.manual_seed(data_seed)
    img = torch.randn(B, 3, 224, 224)
    qst = torch.randint(full['qst_vocab_size'], (B, T))
    qst[:, 0] = 2
    lab = torch.randint(full['ans_vocab_size'], (B,))
    with torch.no_grad():
        ans, qo = ef(img, qst)
        gq, ga = ef.generate(img)
    loss = ef._loss(img, qst, lab)
    named = list(ef.named_parameters())
    grads = torch.autograd.grad(loss, [p for _, p in named], allow_unused=True)
    n_none = sum(g is None for g in grads)
    gsum = {k: checksum(g) for (k, _), g in zip(named, grads) if g is not None}
    gmax = {k: float(g.abs().max()) for (k, _), g in zip(named, grads) if g is not None}

    o_ans, o_qo = O.ef_model_forward(img, qst, P, None, True)
    o_gq, o_ga = O.ef_generate(img, P, None, True, ef.qst_encoder.max_length)
    gen_margins = []
    with torch.no_grad():
        O.lct_generate_tokens(O.img_encoder(img, P, None, frozen=True), P, ef.qst_encoder.max_length, margins=gen_margins)
    gen_margins = torch.stack(gen_margins, 1)                         # [B, 30]
    o_loss = O.ef_loss(img, qst, lab, P, None, True)

    top2a = ans.topk(2, dim=1).values
    top2q = qo.topk(2, dim=2).values
    fx = {'dims': full, 'seed': seed, 'data_seed': data_seed, 'B': B, 'T': T, 'pretrained': True,
          'param_checksums': {k: checksum(v) for k, v in P.items()},
          'image_checksum': checksum(img), 'question': qst, 'label': lab,
          'eval_ans': ans, 'ans_argmax': ans.argmax(1), 'ans_top2_margin': top2a[:, 0] - top2a[:, 1],
          'qst_out_argmax': qo.argmax(2), 'qst_out_top2_margin': top2q[..., 0] - top2q[..., 1],
          'qst_out_checksum': checksum(qo), 'qst_out_absmax': float(qo.abs().max()),
          'qst_out_sample': qo[:, :, ::97].clone(),                   # every 97th vocabulary entry of every position
          'qst_out_rowmax': qo.max(2).values,
          'gen_qst': gq, 'gen_ans': ga, 'gen_top2_margin': gen_margins, 'loss': loss.detach(),
          'grad_checksums': gsum, 'grad_absmax': gmax, 'num_none_grads': n_none}
    torch.save(fx, os.path.join(GOLD, 'ef_full_b4.pt'))
    entry = {'ans_diff': diff(o_ans, ans), 'qst_out_diff': diff(o_qo, qo), 'gen_tokens_equal': bool((o_gq == gq).all()),
             'gen_ans_diff': diff(o_ga, ga), 'loss_diff': diff(o_loss, loss), 'num_none_grads': n_none,
             'ans_top2_margin_min': float((top2a[:, 0] - top2a[:, 1]).min()),
             'qst_out_top2_margin_min': float((top2q[..., 0] - top2q[..., 1]).min()),
             'data_seed': data_seed, 'gen_top2_margin_min': float(gen_margins.min()),
             'ans_absmax': float(ans.abs().max()), 'qst_out_absmax': float(qo.abs().max())}
    mf = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mf))
    manifest['fixtures']['ef_full_b4'] = entry
    with open(mf, 'w') as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(entry, indent=1))
    assert (o_gq == gq).all() and n_none == 36


if __name__ == '__main__':
    main()
