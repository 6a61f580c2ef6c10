"""CPU: the oracle (oracle/vqa_oracle.py) against the golden vectors produced by the reference's
own modules (tests/make_golden.py).  This is what pins the oracle."""
import torch

from oracle import vqa_oracle as O
from _util import load, full_params, rel_err


def test_losses():
    fx = load('losses.pt')
    assert rel_err(O.soft_xent(fx['pred'], fx['target']), fx['softxent']) < 1e-6
    assert rel_err(O.cross_entropy(fx['pred'], fx['label']), fx['ce']) < 1e-6


def test_w_small_eval_train_grads():
    fx = load('w_small.pt')
    P = full_params(fx)
    ev = O.w_model_forward(fx['image'], fx['question'], P)
    assert rel_err(ev, fx['eval_logits']) < 1e-5
    masks = [m.float() for m in fx['train_masks']]
    names = list(fx['grads'].keys())
    Pg = {k: (v.clone().requires_grad_(True) if k in fx['grads'] else v) for k, v in P.items()}
    tr = O.w_model_forward(fx['image'], fx['question'], Pg, masks)
    assert rel_err(tr, fx['train_logits']) < 1e-5
    loss = O.cross_entropy(tr, fx['label'])
    assert rel_err(loss, fx['train_loss']) < 1e-6
    grads = torch.autograd.grad(loss, [Pg[k] for k in names])
    for k, g in zip(names, grads):
        assert rel_err(g, fx['grads'][k]) < 1e-4, k
    assert fx['num_none_grads'] == 36            # frozen VGG: 16 conv w+b, 2 fc w+b
    sl = O.w_soft_loss(fx['image'], fx['question'], fx['label'], fx['pseudo_qst'], fx['pseudo_label'], P, fx['w_lambda'])
    assert rel_err(sl, fx['soft_loss']) < 1e-6


def test_ef_small():
    for name in ('ef_small_frozen.pt', 'ef_small_trainvgg.pt'):
        fx = load(name)
        P = full_params(fx)
        ans, qo = O.ef_model_forward(fx['image'], fx['question'], P, None, fx['pretrained'])
        assert rel_err(ans, fx['eval_ans']) < 1e-5
        assert rel_err(qo, fx['eval_qst_out']) < 1e-5
        gq, ga = O.ef_generate(fx['image'], P, None, fx['pretrained'], fx['gen_qst'].shape[1])
        assert torch.equal(gq, fx['gen_qst'])                      # integer-exact token ids
        assert rel_err(ga, fx['gen_ans']) < 1e-5
        loss = O.ef_loss(fx['image'], fx['question'], fx['label'], P, None, fx['pretrained'])
        assert rel_err(loss, fx['loss']) < 1e-6
        assert fx['num_none_grads'] == (36 if fx['pretrained'] else 0)


def test_w_full_b4_logits_and_argmax():
    """BASELINE config 1: basic_vqa VGG19+LSTM, batch 4, 224x224, 14 tokens, 3129 answers (CPU)."""
    fx = load('w_full_b4.pt')
    from lct_vqa_amd.basic_vqa import models          # product module shells: same init order as the reference
    from _util import checksum, close_sum
    torch.manual_seed(fx['seed'])
    m = models.VqaModel(**fx['dims'])
    P = {k: v.detach() for k, v in m.state_dict().items()}
    assert set(P) == set(fx['param_checksums'])
    for k, cs in fx['param_checksums'].items():
        assert close_sum(checksum(P[k]), cs), k
    torch.manual_seed(fx['data_seed'])
    img = torch.randn(fx['B'], 3, 224, 224)
    assert close_sum(checksum(img), fx['image_checksum'])
    lg = O.w_model_forward(img, fx['question'], P)
    assert rel_err(lg, fx['eval_logits']) < 1e-5
    assert torch.equal(lg.argmax(1), fx['argmax'])


def test_darts_oracle_against_reference_network():
    """oracle/darts_oracle.py vs the reference's pcdarts Network (tests/golden/darts_net.pt)."""
    from oracle import darts_oracle as DO
    fx = load('darts_net.pt')
    P, arch, img = fx['state'], fx['arch'], fx['image']
    assert rel_err(DO.network_forward(img, P, arch, False), fx['eval_out']) < 1e-5
    names = list(fx['param_grads'])
    Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    ag = [a.clone().requires_grad_(True) for a in arch]
    ns = {}
    tr = DO.network_forward(img, Pg, ag, True, ns)
    assert rel_err(tr, fx['train_out']) < 1e-5
    grads = torch.autograd.grad((tr * fx['loss_weight']).sum(), ag + [Pg[n] for n in names])
    for a, r in zip(grads[:4], fx['arch_grads']):
        assert rel_err(a, r) < 1e-4
    for n, a in zip(names, grads[4:]):
        assert rel_err(a, fx['param_grads'][n]) < 1e-4, n
    for k, v in fx['running_after'].items():
        assert rel_err(ns[k], v) < 1e-6, k


def test_ef_full_b4_oracle_against_reference():
    """BASELINE configs[3]'s examiner model at full size (tests/make_golden_full.py ran reference models_lct.VqaModel):
    the oracle on the same seeded parameters and inputs."""
    fx = load('ef_full_b4.pt')
    from lct_vqa_amd.basic_vqa import models_lct, config
    from _util import checksum, close_sum
    config.ARCH_TYPE = 'fixed'
    config.DEVICE = torch.device('cpu')
    torch.manual_seed(fx['seed'])
    m = models_lct.VqaModel(pretrained=True, **fx['dims'])
    P = {k: v.detach() for k, v in m.state_dict().items()}
    assert set(P) == set(fx['param_checksums'])
    for k, cs in fx['param_checksums'].items():
        assert close_sum(checksum(P[k]), cs), k
    torch.manual_seed(fx['data_seed'])
    img = torch.randn(fx['B'], 3, 224, 224)
    assert close_sum(checksum(img), fx['image_checksum'])
    with torch.no_grad():
        ans, qo = O.ef_model_forward(img, fx['question'], P, None, True)
        gq, ga = O.ef_generate(img, P, None, True, fx['gen_qst'].shape[1])
    assert rel_err(ans, fx['eval_ans']) < 1e-5 and torch.equal(ans.argmax(1), fx['ans_argmax'])
    assert torch.equal(qo.argmax(2), fx['qst_out_argmax'])
    assert (qo[:, :, ::97] - fx['qst_out_sample']).abs().max() < 1e-5 * fx['qst_out_absmax']
    assert torch.equal(gq, fx['gen_qst']) and rel_err(ga, fx['gen_ans']) < 1e-5


def test_bf16_storage_emulation_against_resnet_oracle():
    """oracle/resnet_bf16_emu (the checker of the bf16-native trainable trunk): with its roundings switched off it IS the ResNet oracle
    (output and every parameter gradient to 1e-5), which pins its hand-written backward; with them on it shows what bf16 storage costs
    against exact arithmetic at the same point -- the figure the GPU test's fp32 gate (cosine >= 0.85) is derived from."""
    import torch
    from oracle import resnet_oracle as O, resnet_bf16_emu as E
    layers, B, S = (1, 1, 1, 1), 4, 64
    torch.manual_seed(3)
    ref = O.ResNet(layers).train()
    g = torch.Generator().manual_seed(4)
    for name, mod in ref.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.copy_((torch.rand(mod.num_features, generator=g) * 0.5 + 0.75) * (0.25 if name.endswith('bn3') else 1.0))
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    img, dy = torch.randn(B, 3, S, S, generator=g), torch.randn(B, 2048, generator=g)
    yr = ref(img)
    yr.backward(dy)

    def compare():
        feat, grads, _ = E.trunk_forward_backward(sd, layers, img, dy)
        out = ((feat - yr.double()).abs().max() / yr.abs().max()).item()
        l2 = [((grads[n] - p.grad.double()).norm() / p.grad.double().norm()).item() for n, p in ref.named_parameters()]
        cos = [(grads[n].flatten() @ p.grad.double().flatten() / grads[n].norm() / p.grad.double().norm()).item() for n, p in ref.named_parameters()]
        return out, max(l2), min(cos)
    rounding = E.r
    try:
        E.r = lambda t: t.double()
        out, l2, cos = compare()
        assert out < 1e-5 and l2 < 1e-5, (out, l2)
    finally:
        E.r = rounding
    out, l2, cos = compare()
    assert out < 3e-2 and 0.02 < l2 < 0.5 and cos > 0.9, (out, l2, cos)
