"""Frozen ResNet-152 trunk (lct_vqa_amd/resnet.py, csrc/resnet.hip) for BASELINE.json configs[2] against the CPU restatement of
torchvision's published architecture (oracle/resnet_oracle.py).  The reference has no ResNet: parity is
unpinned by the reference; the gate is the north_star's 1e-3 relative on fp32 outputs."""
import pytest
import torch

from _util import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _randomise_bn(m, seed):
    g = torch.Generator().manual_seed(seed)
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_((torch.rand(mod.num_features, generator=g) * 0.5 + 0.75) * (0.25 if name.endswith('bn3') else 1.0))
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)


def test_stem_fold_and_pool_kernels(hip_lib):
    import torch.nn.functional as F
    lib = hip_lib
    torch.manual_seed(1)
    B, H, W = 2, 38, 50
    img = torch.randn(B, 3, H, W, device=DEV)
    w = torch.randn(64, 3, 7, 7, device=DEV) * 0.1
    gamma, beta = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1
    mean, var = torch.randn(64, device=DEV) * 0.1, torch.rand(64, device=DEV) + 0.5
    wf, shift = torch.empty_like(w), torch.empty(64, device=DEV)
    lib('fold_bn', w, gamma, beta, mean, var, 1e-5, 64, 147, wf, shift)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, 64), device=DEV)
    lib('conv7x7_s2_stem', img, wf, shift, y, B, H, W, 1)
    ref = F.relu(F.batch_norm(F.conv2d(img, w, None, stride=2, padding=3), mean, var, gamma, beta, False, 0.0, 1e-5))
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-5
    x = torch.randn(B, 5, 7, 96, device=DEV)
    out = torch.empty((B, 96), device=DEV)
    lib('global_avgpool_nhwc', x, out, B, 35, 96)
    assert rel_err(out, x.mean(dim=(1, 2))) < 1e-6


@pytest.mark.parametrize('size', [64, 96, 224])
def test_resnet152_trunk_vs_oracle(hip_lib, size):
    from oracle import resnet_oracle                            # checker only
    from lct_vqa_amd.resnet import ResNet152Trunk
    torch.manual_seed(3)
    ref = resnet_oracle.resnet152().eval()
    _randomise_bn(ref, 4)
    net = ResNet152Trunk()
    assert set(net.state_dict()) == set(ref.state_dict())       # torchvision's names: its checkpoints load
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV).train()                                   # a frozen trunk uses running statistics in either mode
    img = torch.randn(2, 3, size, size)
    with torch.no_grad():
        want = ref(img)
        got = net(img.to(DEV))
        again = net(img.to(DEV))
    assert got.shape == (2, 2048) and torch.equal(got, again)
    assert rel_err(got, want) < 1e-3
    for p in net.parameters():
        p.requires_grad_(True)
    with pytest.raises(NotImplementedError):
        net(img.to(DEV))                                        # not silently wrong: no gradients through this trunk


def test_resnet_gru_model_train_step(hip_lib):
    """configs[2]'s model shape end to end: logits against the pieces' torch formulas, gradients reach exactly the head."""
    from oracle import resnet_oracle
    from lct_vqa_amd import ops
    from lct_vqa_amd.basic_vqa import config, models_ext
    config.DEVICE = torch.device(DEV)
    torch.manual_seed(5)
    dims = dict(embed_size=32, qst_vocab_size=50, ans_vocab_size=40, word_embed_size=16, num_layers=1, hidden_size=32)
    m = models_ext.VqaModelResNetGRU(**dims)
    _randomise_bn(m.img_encoder.model, 6)
    ref_trunk = resnet_oracle.resnet152().eval()
    ref_trunk.load_state_dict(m.img_encoder.model.state_dict())
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    img = torch.randn(3, 3, 64, 64)
    qst = torch.randint(50, (3, 9)); qst[:, 0] = 2
    lab = torch.randint(40, (3,))
    logits = m(img.to(DEV), qst.to(DEV))
    with torch.no_grad():                                       # the same model from torch's own modules
        f = ref_trunk(img) @ P['img_encoder.fc.weight'].T + P['img_encoder.fc.bias']
        f = f / f.norm(p=2, dim=1, keepdim=True)
        gru = torch.nn.GRU(16, 32, 1)
        gru.load_state_dict({k.split('gru.')[1]: v for k, v in P.items() if '.gru.' in k})
        e = torch.tanh(torch.nn.functional.embedding(qst, P['qst_encoder.word2vec.weight'])).transpose(0, 1)
        _, h = gru(e)
        q = torch.tanh(h[0]) @ P['qst_encoder.fc.weight'].T + P['qst_encoder.fc.bias']
        z = torch.tanh(f * q)
        z = torch.tanh(z @ P['fc1.weight'].T + P['fc1.bias'])
        want = z @ P['fc2.weight'].T + P['fc2.bias']
    assert rel_err(logits, want) < 1e-3 and torch.equal(logits.argmax(1).cpu(), want.argmax(1))
    loss = ops.cross_entropy(logits, lab.to(DEV))
    grads = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
    none = {n for (n, _), g in zip(m.named_parameters(), grads) if g is None}
    assert none == {n for n, _ in m.named_parameters() if n.startswith('img_encoder.model.')}
