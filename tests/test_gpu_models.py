"""GPU parity of the whole models through the plugin surface against the golden vectors that the
REFERENCE's own modules produced (tests/golden), and against the oracle on fresh seeded inputs."""
import pytest
import torch

from _util import load, full_params, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def pkg(hip_lib):
    from lct_vqa_amd import ops
    from lct_vqa_amd.basic_vqa import models, models_lct, config
    config.DEVICE = torch.device(DEV)
    return ops, models, models_lct, config


def _build(cls, fx, **kw):
    m = cls(**fx['dims'], **kw)
    m.load_state_dict(full_params(fx))
    return m.to(DEV)


def test_w_small_against_reference_goldens(pkg):
    ops, models, _, _ = pkg
    fx = load('w_small.pt')
    m = _build(models.VqaModel, fx)
    img, qst, lab = fx['image'].to(DEV), fx['question'].to(DEV), fx['label'].to(DEV)
    m.eval()
    with torch.no_grad():
        ev = m(img, qst)
    assert rel_err(ev, fx['eval_logits']) < 1e-3
    assert torch.equal(ev.argmax(1).cpu(), fx['eval_logits'].argmax(1))
    m.train()
    with ops.inject_masks(fx['train_masks']):
        tr = m(img, qst)
        loss = ops.cross_entropy(tr, lab)
    assert rel_err(tr, fx['train_logits']) < 1e-3 and rel_err(loss, fx['train_loss']) < 1e-4
    grads = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
    assert sum(g is None for g in grads) == fx['num_none_grads'] == 36
    for (k, _), g in zip(m.named_parameters(), grads):
        if g is not None:
            assert rel_err(g, fx['grads'][k]) < 1e-3, k
    m.eval()
    with torch.no_grad():
        sl = m._soft_loss(img, qst, lab, fx['pseudo_qst'].to(DEV), fx['pseudo_label'].to(DEV))
        # the same with the frozen trunk's conv features evaluated by the caller (bench.py --workload lct runs them on
        # a second stream)
        sl2 = m._soft_loss(img, qst, lab, fx['pseudo_qst'].to(DEV), fx['pseudo_label'].to(DEV), m.img_encoder.conv_features(img))
    assert rel_err(sl, fx['soft_loss']) < 1e-4 and torch.equal(sl, sl2)


def test_ef_small_against_reference_goldens(pkg):
    ops, _, models_lct, config = pkg
    fx = load('ef_small_frozen.pt')
    config.ARCH_TYPE = 'fixed'
    m = _build(models_lct.VqaModel, fx, pretrained=True).eval()
    img, qst, lab = fx['image'].to(DEV), fx['question'].to(DEV), fx['label'].to(DEV)
    with torch.no_grad():
        ans, qo = m(img, qst)
        gq, ga = m.generate(img)
    assert rel_err(ans, fx['eval_ans']) < 1e-3 and rel_err(qo, fx['eval_qst_out']) < 1e-3
    assert torch.equal(gq.cpu(), fx['gen_qst'])          # greedy token ids: bit-exact
    assert rel_err(ga, fx['gen_ans']) < 1e-3
    # a driver that owns the loop may evaluate the frozen trunk's conv features once per batch and share them between
    # stage 1 (forward / _loss) and stage 2 (generate): same goldens through that path
    with torch.no_grad():
        conv = m.img_encoder.conv_features(img)
        ans_s, qo_s = m(img, qst, conv)
        gq_s, ga_s = m.generate(img, conv)
    assert rel_err(ans_s, fx['eval_ans']) < 1e-3 and rel_err(qo_s, fx['eval_qst_out']) < 1e-3
    assert torch.equal(gq_s.cpu(), fx['gen_qst']) and rel_err(ga_s, fx['gen_ans']) < 1e-3
    assert rel_err(m._loss(img, qst, lab, conv), fx['loss']) < 1e-4
    loss = m._loss(img, qst, lab)
    assert rel_err(loss, fx['loss']) < 1e-4
    grads = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
    assert sum(g is None for g in grads) == 36
    for (k, _), g in zip(m.named_parameters(), grads):
        if g is not None:
            assert rel_err(g, fx['grads'][k]) < 1e-3, k
    # gradient through generated answers reaches everything except the token head (architect_lct.py:77: 2 Nones
    # among trainable tensors + the 36 frozen ones)
    wq, wa = m.generate(img)
    g2 = torch.autograd.grad(ops.soft_xent(wa, torch.softmax(fx['gen_ans'].to(DEV), 1)), list(m.parameters()), allow_unused=True)
    none_names = [k for (k, _), g in zip(m.named_parameters(), g2) if g is None and '.model.' not in k]
    assert none_names == [] or set(none_names) <= {'qst_encoder.fc2.weight', 'qst_encoder.fc2.bias'}


def test_w_full_b4_logits_argmax(pkg):
    """BASELINE config 1 shapes on the GPU: B=4, 224x224, T=14, A=3129, QV=17858, vs the reference's logits."""
    ops, models, _, _ = pkg
    fx = load('w_full_b4.pt')
    torch.manual_seed(fx['seed'])
    m = models.VqaModel(**fx['dims']).to(DEV).eval()
    torch.manual_seed(fx['data_seed'])
    img = torch.randn(fx['B'], 3, 224, 224).to(DEV)
    with torch.no_grad():
        lg = m(img, fx['question'].to(DEV))
    assert rel_err(lg, fx['eval_logits']) < 1e-3
    assert torch.equal(lg.argmax(1).cpu(), fx['argmax'])


def test_train_step_matches_oracle(pkg):
    """One full W train step (fwd, CE, bwd, clip 5, Adam) vs the oracle's, injected masks."""
    ops, models, _, _ = pkg
    from oracle import vqa_oracle as O
    from lct_vqa_amd.optim import FusedAdam
    fx = load('w_small.pt')
    m = _build(models.VqaModel, fx).train()
    P = full_params(fx)
    img, qst, lab = fx['image'], fx['question'], fx['label']
    opt = FusedAdam(m.trainable_parameters(), lr=1e-3, flatten=True)
    state = {}
    for it in range(2):
        masks = [mk.float() for mk in fx['train_masks']]
        loss_o, _, P, _ = O.w_train_step(img, qst, lab, P, state, masks)
        opt.zero_grad()
        with ops.inject_masks(fx['train_masks']):
            loss = ops.cross_entropy(m(img.to(DEV), qst.to(DEV)), lab.to(DEV))
        loss.backward()
        opt.step(max_grad_norm=5)
        assert rel_err(loss, loss_o) < 1e-4
    for k, v in m.state_dict().items():
        assert rel_err(v, P[k]) < 1e-3, k


def test_ef_small_trainable_vgg_against_reference_goldens(pkg):
    """EF model with pretrained=False (trainable VGG19, reference models_lct.py:62-63): loss, every small gradient,
    and sum / abs-sum checksums of the VGG gradients produced by the reference."""
    ops, _, models_lct, config = pkg
    from _util import checksum
    fx = load('ef_small_trainvgg.pt')
    config.ARCH_TYPE = 'fixed'
    m = _build(models_lct.VqaModel, fx, pretrained=False).eval()
    img, qst, lab = fx['image'].to(DEV), fx['question'].to(DEV), fx['label'].to(DEV)
    loss = m._loss(img, qst, lab)
    assert rel_err(loss, fx['loss']) < 1e-4
    grads = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
    assert sum(g is None for g in grads) == fx['num_none_grads'] == 0
    for (k, _), g in zip(m.named_parameters(), grads):
        if k in fx['grads']:
            assert rel_err(g, fx['grads'][k]) < 1e-3, k
        else:
            s, a = checksum(g)
            rs, ra = fx['vgg_grad_checksums'][k]
            assert abs(a - ra) <= 2e-3 * max(ra, 1e-12), (k, a, ra)
            assert abs(s - rs) <= 2e-3 * max(ra, 1e-12), (k, s, rs)


def test_checkpoint_wire_compat_with_torch_adam(pkg, tmp_path):
    """Reference checkpoints (experiment.py:398-417) hold {'w_model': state_dict, 'w_opt': Adam.state_dict(), ...}.
    A torch.optim.Adam state written by the reference must resume in FusedAdam (per-tensor and flat arena), and a
    FusedAdam checkpoint must load back into torch.optim.Adam."""
    ops, models, _, _ = pkg
    from lct_vqa_amd.optim import FusedAdam
    fx = load('w_small.pt')
    m = _build(models.VqaModel, fx).eval()
    img, qst, lab = fx['image'].to(DEV), fx['question'].to(DEV), fx['label'].to(DEV)
    ref_opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for _ in range(2):
        ref_opt.zero_grad()
        ops.cross_entropy(m(img, qst), lab).backward()
        ref_opt.step()
    ck = {'w_model': m.state_dict(), 'w_opt': ref_opt.state_dict()}
    torch.save(ck, tmp_path / 'w_model.pt')
    ck = torch.load(tmp_path / 'w_model.pt', map_location=DEV, weights_only=False)
    results = []
    for flatten in (False, True):
        m2 = _build(models.VqaModel, fx).eval()
        m2.load_state_dict(ck['w_model'])
        opt = FusedAdam(list(m2.parameters()), lr=1e-3, flatten=flatten)
        opt.load_state_dict(ck['w_opt'])
        opt.zero_grad()
        ops.cross_entropy(m2(img, qst), lab).backward()
        opt.step()
        results.append(m2.fc2.weight.detach().clone())
        sd = opt.state_dict()
        assert len(sd['state']) >= 13 and all(set(v) >= {'step', 'exp_avg', 'exp_avg_sq'} for v in sd['state'].values())
    # continue the torch optimizer one more step: all three must agree
    ref_opt.zero_grad()
    ops.cross_entropy(m(img, qst), lab).backward()
    ref_opt.step()
    for r in results:
        assert rel_err(r, m.fc2.weight) < 1e-5
    t_opt = torch.optim.Adam(m2.parameters(), lr=1e-3)
    t_opt.load_state_dict(sd)                      # FusedAdam(flat) -> torch.optim.Adam


def test_gru_variant_model_vs_torch(pkg):
    """VGG19 + GRU variant (BASELINE.json configs[2] recurrent half; not in the reference): logits and gradients vs
    the same arithmetic in plain torch (nn.GRU) on CPU, frozen trunk output shared."""
    ops, models, _, _ = pkg
    from lct_vqa_amd.basic_vqa import models_ext
    torch.manual_seed(5)
    m = models_ext.VqaModelGRU(32, 50, 40, 16, 1, 32)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    md = m.to(DEV).eval()
    img = torch.randn(3, 3, 32, 32)
    qst = torch.randint(50, (3, 9))
    lab = torch.randint(40, (3,))
    lg = md(img.to(DEV), qst.to(DEV))
    loss = ops.cross_entropy(lg, lab.to(DEV))
    names = [n for n, _ in md.named_parameters() if not n.startswith('img_encoder.model.')]
    grads = torch.autograd.grad(loss, [dict(md.named_parameters())[n] for n in names])
    # torch restatement on CPU
    from oracle import vqa_oracle as O
    P = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    img_f = O.img_encoder(img, P, None, True)
    gru = torch.nn.GRU(16, 32, 1)
    v = torch.tanh(P['qst_encoder.word2vec.weight'][qst]).transpose(0, 1)
    _, h = torch._VF.gru(v, torch.zeros(1, 3, 32), [P['qst_encoder.gru.weight_ih_l0'], P['qst_encoder.gru.weight_hh_l0'],
                                                     P['qst_encoder.gru.bias_ih_l0'], P['qst_encoder.gru.bias_hh_l0']],
                         True, 1, 0.0, False, False, False)
    qf = torch.nn.functional.linear(torch.tanh(h[0]), P['qst_encoder.fc.weight'], P['qst_encoder.fc.bias'])
    ref = O.fusion_head(img_f, qf, P, None)
    rloss = O.cross_entropy(ref, lab)
    rg = torch.autograd.grad(rloss, [P[n] for n in names])
    assert rel_err(lg, ref) < 1e-3 and torch.equal(lg.argmax(1).cpu(), ref.argmax(1))
    for n, a, r in zip(names, grads, rg):
        assert rel_err(a, r) < 1e-3, n


def test_eval_path_counters_vs_oracle(pkg):
    """Validation counters (SURVEY f-2; reference experiment.py:314-324, misc.py:42-58): fused argmax + '<unk>' masking +
    multiple-choice match on the device vs the oracle's restatement; includes ties (first index wins, like torch.max),
    predictions equal to '<unk>', and an epoch of several ragged batches."""
    from oracle import vqa_oracle as O
    from lct_vqa_amd.basic_vqa import misc
    g = torch.Generator().manual_seed(11)
    A, K, unk = 37, 10, 1
    ev = misc.Evaluator(unk, torch.device(DEV))
    tot = [0.0, 0, 0]
    nb = 0
    for B in (1, 7, 64, 5):
        out = torch.randn(B, A, generator=g)
        out[0, 3] = out[0, 9] = out[0].max() + 1.0                      # tie: index 3 must win
        if B > 2:
            out[2, unk] = out[2].max() + 2.0                            # predicts '<unk>'
        label = torch.randint(0, A, (B,), generator=g)
        mc = [torch.randint(0, A, (B,), generator=g) for _ in range(K)]
        mc[4][0] = 3
        if B > 2:
            mc[1][2] = unk                                              # '<unk>' is among the choices: Exp1 hit, Exp2 miss
        l, c1, c2 = O.val_counters(out.clone(), label, mc, unk)
        tot[0] += l; tot[1] += c1; tot[2] += c2; nb += 1
        pred = ev.update(out.to(DEV), label.to(DEV), mc)
        assert torch.equal(pred.cpu(), out.argmax(1))
        assert misc.num_correct(pred, mc) == O.num_correct(out.argmax(1), mc)
    r = ev.result()
    assert (r['corr_exp1'], r['corr_exp2']) == (tot[1], tot[2])
    assert r['corr_exp1'] > r['corr_exp2']
    assert abs(r['loss'] - tot[0] / nb) < 1e-5
    # question accuracy buckets
    qp = torch.randn(6, 9, 21, generator=g)
    q = torch.randint(0, 21, (6, 9), generator=g)
    q[0, 1:] = qp[0].argmax(1)[:-1]                                     # one fully correct question
    assert misc.num_correct_qst(qp.to(DEV), q.to(DEV)) == O.num_correct_qst(qp, q)


def test_ragged_and_empty_batches(pkg):
    """Batch sizes that are not multiples of any tile (1, 3, 5: the last, short batch of an epoch) give the same per-sample
    logits as a larger batch; an EMPTY batch is refused loudly by the C ABI (shape check) instead of launching empty grids."""
    ops, models, _, _ = pkg
    torch.manual_seed(2)
    m = models.VqaModel(embed_size=64, qst_vocab_size=50, ans_vocab_size=20, word_embed_size=16, num_layers=1,
                        hidden_size=32).to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    img = torch.randn(5, 3, 64, 64, generator=g).to(DEV)
    qst = torch.randint(0, 50, (5, 7), generator=g).to(DEV)
    with torch.no_grad():
        full = m(img, qst)
        for B in (1, 3):
            assert rel_err(m(img[:B].contiguous(), qst[:B].contiguous()), full[:B]) < 1e-5
        with pytest.raises(RuntimeError):
            m(img[:0].contiguous(), qst[:0].contiguous())


@pytest.mark.parametrize('H,We,QV,B,L', [(512, 300, 1000, 37, 9), (32, 16, 50, 5, 14), (64, 20, 77, 33, 6)])
def test_greedy_decode_in_c_equals_the_token_loop(pkg, H, We, QV, B, L):
    """lctvqa_lstm_decode_greedy (one C call for the whole generate loop) against the per-token path it replaces
    (embed -> lstm -> tanh -> token head -> argmax through the autograd wrappers): identical token ids."""
    ops, _, models_lct, config = pkg
    config.ARCH_TYPE = 'fixed'
    torch.manual_seed(H + L)
    enc = models_lct.QstEncoder(QV, We, H, 1, H).to(DEV).eval()
    enc.max_length = L
    feat = torch.randn(B, H, device=DEV) * 0.5
    with torch.no_grad():
        enc.deterministic = True
        fast = enc.generate(feat)
        # the loop: force the fallback branch by asking the stochastic path for its deterministic sampler
        h = c = feat
        word = ops.embed(torch.full((B, 1), 2, dtype=torch.long, device=DEV), enc.word2vec.weight, True)
        params = (enc.lstm.weight_ih_l0, enc.lstm.weight_hh_l0, enc.lstm.bias_ih_l0, enc.lstm.bias_hh_l0)
        slow = torch.zeros(B, L, dtype=torch.long, device=DEV)
        for t in range(L):
            out, h, c = ops.lstm(word, *params, h, c)
            pred = ops.argmax_rows(ops.linear(ops.TanhFn.apply(out), enc.fc2.weight, enc.fc2.bias).reshape(B, -1)).reshape(B, 1)
            word = ops.embed(pred, enc.word2vec.weight, False)
            slow[:, t] = pred[:, 0]
    assert fast.dtype == torch.long and tuple(fast.shape) == (B, L)
    assert torch.equal(fast, slow)


def test_stochastic_sample_follows_the_reference_distribution(hip_lib):
    """QstEncoder.sample with deterministic=False (reference models_lct.py:159-170: WeightedRandomSampler on softmax(prob / temperature),
    one draw per row): the HIP softmax feeds a seeded torch.multinomial; 40 000 independent rows of the same logits must reproduce the
    oracle's distribution softmax(logits / T) within 5 standard deviations per token, and the deterministic branch returns the argmax."""
    from lct_vqa_amd.basic_vqa import models_lct
    enc = models_lct.QstEncoder(qst_vocab_size=12, word_embed_size=4, embed_size=8, num_layers=1, hidden_size=8).to(DEV)
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(12, generator=g) * 1.5
    B = 40000
    prob = logits.expand(B, 1, 12).contiguous().to(DEV)
    for T in (1.0, 0.5):
        enc.deterministic, enc.temperature = False, T
        torch.manual_seed(7)
        ids = enc.sample(prob)
        assert ids.shape == (B, 1) and ids.dtype == torch.int64
        want = torch.softmax(logits.double() / T, 0)
        freq = torch.bincount(ids.flatten().cpu(), minlength=12).double() / B
        sigma = torch.sqrt(want * (1 - want) / B)
        assert ((freq - want).abs() <= 5 * sigma + 1e-9).all(), (T, freq, want)
        torch.manual_seed(7)
        assert torch.equal(ids, enc.sample(prob))                      # the same seed draws the same tokens
    enc.deterministic = True
    assert torch.equal(enc.sample(prob).flatten().cpu(), torch.full((B,), int(logits.argmax())))
