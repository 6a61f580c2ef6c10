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


@pytest.mark.parametrize('B,H,W', [(2, 38, 50), (3, 224, 224), (1, 8, 256), (2, 30, 8)])
def test_stem_on_matrix_cores_bf16(hip_lib, B, H, W):
    """conv7x7/s2 stem as an implicit GEMM on bf16 MFMA: against torch on the SAME bf16-rounded image and weights (fp32
    accumulation; the output's own rounding to bf16 is the tolerance), ragged widths, bottom rows of a 4-row group, borders."""
    import torch.nn.functional as F
    lib = hip_lib
    assert lib.query('conv7x7_s2_stem_bf16_mfma_supported', B, H, W)
    g = torch.Generator().manual_seed(H + W)
    img = torch.randn(B, 3, H, W, generator=g).to(DEV)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(DEV)
    shift = torch.randn(64, generator=g).to(DEV)
    Ho, Wo = H // 2, W // 2
    for relu in (1, 0):
        y = torch.full((B, Ho, Wo, 64), float('nan'), dtype=torch.bfloat16, device=DEV)
        lib('conv7x7_s2_stem_bf16_mfma', img, w, shift, y, B, H, W, relu)
        ref = F.conv2d(img.bfloat16().double(), w.bfloat16().double(), shift.double(), stride=2, padding=3)
        if relu:
            ref = F.relu(ref)
        got = y.double().permute(0, 3, 1, 2)
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()


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


def test_bn_act_and_conv3x3_functions_vs_torch(hip_lib):
    """The bottleneck's autograd nodes one by one against torch autograd: wide BatchNorm (+ residual + ReLU) in both modes, the
    3x3 convolution on a tiled map and on a small map (shifted-GEMM path), the stem, the global average pool."""
    import torch.nn.functional as F
    from lct_vqa_amd import resnet_ops as R
    torch.manual_seed(11)
    for C, H, training, with_res, relu in ((512, 5, True, True, True), (64, 12, True, False, True), (2048, 2, False, True, True),
                                            (256, 7, True, False, False)):
        x = torch.randn(3, H, H, C, device=DEV, requires_grad=True)
        res = torch.randn(3, H, H, C, device=DEV, requires_grad=True) if with_res else None
        bn = torch.nn.BatchNorm2d(C).to(DEV).train(training)
        bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
        ref_bn = torch.nn.BatchNorm2d(C).to(DEV).train(training)
        ref_bn.load_state_dict(bn.state_dict())
        y = R.BnActFn.apply(x, bn.weight, bn.bias, res, bn, relu)
        xr = x.detach().clone().requires_grad_()
        rr = res.detach().clone().requires_grad_() if with_res else None
        yr = ref_bn(xr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        if with_res:
            yr = yr + rr
        if relu:
            yr = F.relu(yr)
        assert rel_err(y, yr) < 1e-5
        gy = torch.randn_like(y)
        got = torch.autograd.grad(y, [x, bn.weight, bn.bias] + ([res] if with_res else []), gy)
        want = torch.autograd.grad(yr, [xr, ref_bn.weight, ref_bn.bias] + ([rr] if with_res else []), gy)
        for a, b in zip(got, want):
            assert rel_err(a, b) < 2e-4, (C, H, training)
        assert rel_err(bn.running_mean, ref_bn.running_mean) < 1e-5 and rel_err(bn.running_var, ref_bn.running_var) < 1e-5
        assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)
    for Cin, Cout, H in ((64, 64, 16), (128, 128, 7), (512, 512, 3), (256, 256, 14)):
        x = torch.randn(2, H, H, Cin, device=DEV, requires_grad=True)
        w = (torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05).requires_grad_()
        y = R.Conv3x3Fn.apply(x, w)
        xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
        yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1)
        assert rel_err(y, yr) < 2e-5, (Cin, H)
        gy = torch.randn_like(y)
        for a, b in zip(torch.autograd.grad(y, [x, w], gy), torch.autograd.grad(yr, [xr, wr], gy)):
            assert rel_err(a, b) < 1e-4, (Cin, H)
    img = torch.randn(2, 3, 36, 44, device=DEV)
    w = (torch.randn(64, 3, 7, 7, device=DEV) * 0.1).requires_grad_()
    y = R.StemFn.apply(img, w)
    wr = w.detach().clone().requires_grad_()
    yr = F.conv2d(img, wr, stride=2, padding=3).permute(0, 2, 3, 1)
    gy = torch.randn_like(y)
    assert rel_err(y, yr) < 1e-5
    assert rel_err(torch.autograd.grad(y, [w], gy)[0], torch.autograd.grad(yr, [wr], gy)[0]) < 1e-4
    x = torch.randn(3, 4, 5, 32, device=DEV, requires_grad=True)
    y = R.GlobalAvgPoolFn.apply(x)
    assert rel_err(y, x.mean(dim=(1, 2))) < 1e-6
    assert rel_err(torch.autograd.grad(y.sum() * 2, [x])[0], torch.full_like(x, 2.0 / 20)) < 1e-6


def test_trainable_resnet152_vs_oracle(hip_lib):
    """The trainable trunk (train-mode BatchNorm: batch statistics) against torch autograd on the restated architecture: output,
    running statistics and the gradients of parameters from every stage."""
    from oracle import resnet_oracle
    from lct_vqa_amd.resnet import ResNet152Trunk
    torch.manual_seed(7)
    ref = resnet_oracle.resnet152().train()
    _randomise_bn(ref, 8)
    net = ResNet152Trunk()
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV).train()
    img = torch.randn(4, 3, 64, 64)
    want = ref(img)
    got = net(img.to(DEV))
    assert rel_err(got, want) < 2e-3
    names = ['conv1.weight', 'bn1.weight', 'layer1.0.conv2.weight', 'layer1.0.downsample.0.weight', 'layer2.0.conv2.weight',
             'layer2.3.bn2.bias', 'layer3.0.downsample.1.weight', 'layer3.20.conv1.weight', 'layer4.0.conv2.weight', 'layer4.2.conv3.weight',
             'layer4.2.bn3.weight']
    gy = torch.randn(4, 2048)
    rp, npar = dict(ref.named_parameters()), dict(net.named_parameters())
    gw = torch.autograd.grad(want, [rp[n] for n in names], gy)
    gg = torch.autograd.grad(got, [npar[n] for n in names], gy.to(DEV))
    for n, a, b in zip(names, gg, gw):
        assert rel_err(a, b) < 3e-2, (n, rel_err(a, b))
    sd, rsd = net.state_dict(), ref.state_dict()
    for k in ('bn1.running_mean', 'layer2.0.bn2.running_var', 'layer4.2.bn3.running_mean', 'layer3.5.bn1.num_batches_tracked'):
        assert rel_err(sd[k].float(), rsd[k].float()) < 1e-3, k


def test_gemm_bf16_vs_torch(hip_lib):
    lib = hip_lib
    torch.manual_seed(13)
    for M, N, K, act, with_add, out_bf16 in ((1000, 128, 64, 1, True, 1), (257, 64, 256, 0, False, 1), (3136, 256, 64, 1, False, 1),
                                              (500, 2048, 512, 1, True, 1), (640, 512, 512, 0, True, 0)):
        A = torch.randn(M, K, device=DEV).bfloat16()
        Bw = (torch.randn(N, K, device=DEV) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=DEV) * 0.1
        add = torch.randn(M, N, device=DEV)
        add = add.bfloat16() if out_bf16 else add
        C = torch.full((M, N), float('nan'), device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        lib('gemm_bf16', A, K, Bw, K, C, N, M, N, K, bias, add if with_add else None, N, act, out_bf16)
        ref = A.float() @ Bw.float().T + bias + (add.float() if with_add else 0)
        if act:
            ref = torch.relu(ref)
        tol = 6e-3 if out_bf16 else 1e-5
        assert (C.float() - ref).abs().max().item() <= tol * ref.abs().max().item(), (M, N, K)
    C0 = torch.randn(300, 128, device=DEV)                      # fp32 accumulation over two calls (addend aliases the output)
    A = torch.randn(300, 96, device=DEV).bfloat16(); Bw = torch.randn(128, 96, device=DEV).bfloat16()
    C = C0.clone()
    lib('gemm_bf16', A, 96, Bw, 96, C, 128, 300, 128, 96, None, C, 128, 0, 0)
    assert rel_err(C, C0 + A.float() @ Bw.float().T) < 1e-5


@pytest.mark.parametrize('size', [64, 224])
def test_resnet152_trunk_bf16_vs_oracle(hip_lib, size):
    """The frozen trunk in bf16 against the fp32 oracle: ~150 layers of bf16 activations; tolerance 5e-2 of the largest feature
    (measured ~1e-2), the feature ranking of each image essentially unchanged (correlation > 0.999)."""
    from oracle import resnet_oracle
    from lct_vqa_amd import ops
    from lct_vqa_amd.resnet import ResNet152Trunk
    torch.manual_seed(3)
    ref = resnet_oracle.resnet152().eval()
    _randomise_bn(ref, 4)
    net = ResNet152Trunk()
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV).eval()
    img = torch.randn(2, 3, size, size)
    with torch.no_grad():
        want = ref(img)
    prev = ops.set_compute_dtype('bf16')
    try:
        with torch.no_grad():
            got = net(img.to(DEV))
    finally:
        ops.set_compute_dtype(prev)
    assert got.dtype == torch.float32 and got.shape == want.shape
    assert rel_err(got, want) < 5e-2
    g, w = got.cpu().double(), want.double()
    for i in range(2):
        assert torch.corrcoef(torch.stack([g[i], w[i]]))[0, 1].item() > 0.999


def test_trainable_trunk_under_no_grad_sees_the_optimizer_and_batchnorm_updates(hip_lib):
    """Regression: train -> no_grad evaluation -> train -> no_grad evaluation of a TRAINABLE trunk, the order of
    the reference's epoch loop (experiment.py:150-176: train(), then val() under no_grad).  The no_grad forward folds BatchNorm
    into the convolutions; FusedAdam's kernel and the BatchNorm kernels change weights and running statistics through raw
    pointers (no version counter moves), so a folded copy kept from the first evaluation would be served for ever.  The second
    evaluation must equal the oracle on the UPDATED state_dict -- and differ from the first.  (``cache_folded_weights`` is the
    opt-in for trunks nothing ever writes to; it is off here, as it is by default.)"""
    from oracle import resnet_oracle
    from lct_vqa_amd.resnet import ResNet152Trunk
    from lct_vqa_amd.optim import FusedAdam
    torch.manual_seed(17)
    net = ResNet152Trunk()
    _randomise_bn(net, 18)
    net = net.to(DEV)
    img = torch.randn(4, 3, 64, 64)
    ev = torch.randn(2, 3, 64, 64)
    opt = FusedAdam(list(net.parameters()), lr=1e-2, flatten=True)

    def evaluate():
        net.eval()
        with torch.no_grad():
            got = net(ev.to(DEV))
        ref = resnet_oracle.resnet152().eval()
        ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
        with torch.no_grad():
            want = ref(ev)
        return got, want

    def train_step():
        net.train()
        opt.zero_grad()
        net(img.to(DEV)).square().mean().backward()
        opt.step(max_grad_norm=5.0)

    got0, want0 = evaluate()
    assert rel_err(got0, want0) < 2e-3
    train_step()
    got1, want1 = evaluate()
    assert rel_err(want1, want0) > 1e-2                            # the step and the running statistics did move the function
    assert rel_err(got1, want1) < 2e-3                             # ... and the no_grad path follows
    train_step()
    got2, want2 = evaluate()
    assert rel_err(got2, want2) < 2e-3
