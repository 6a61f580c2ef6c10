"""The heads under the bf16 compute dtype (ops.set_compute_dtype('bf16'): BASELINE.json's bf16 configurations; the reference is
fp32 only).  Every nn.Linear product -- forward, dX, dW -- multiplies bf16 copies of its fp32 operands with fp32 accumulation
(csrc/gemm_bf16.hip gemm_bf16x_kernel), and the LCT token head is fused with its cross entropy (LinearXentFn).  Gates: each kernel
against float64 products of the SAME bf16-rounded operands (what remains is fp32 accumulation order: 1e-5), and model-level
results against the fp32 path / the reference's fp32 goldens with the stated bf16 tolerances."""
import pytest
import torch

from _util import load, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def r16(t):
    """round to bf16 and back, in float64"""
    return t.bfloat16().double()


@pytest.fixture
def bf16_mode(hip_lib):
    from lct_vqa_amd import ops
    prev = ops.set_compute_dtype('bf16')
    yield ops
    ops.set_compute_dtype(prev)


@pytest.mark.parametrize('R,C,rp,cp', [(5, 7, 8, 8), (300, 17858, 300, 17888), (64, 64, 64, 64), (33, 100, 128, 128)])
def test_cvt_pad_bf16(hip_lib, R, C, rp, cp):
    torch.manual_seed(R + C)
    big = torch.randn(R, C + 3, device=DEV)
    src = big[:, :C]                                             # row stride C + 3: not 16-byte aligned rows
    dst = torch.full((rp, cp), 7.0, dtype=torch.bfloat16, device=DEV)
    hip_lib('cvt_pad_bf16', src, src.stride(0), R, C, dst, cp, rp, cp, 0)
    want = torch.zeros(rp, cp, dtype=torch.bfloat16, device=DEV)
    want[:R, :C] = src.bfloat16()
    assert torch.equal(dst, want)
    rp64, cp64 = -(-R // 64) * 64, -(-C // 64) * 64
    dstT = torch.full((cp64, rp64), 7.0, dtype=torch.bfloat16, device=DEV)
    hip_lib('cvt_pad_bf16', src, src.stride(0), R, C, dstT, rp64, rp64, cp64, 1)
    wantT = torch.zeros(cp64, rp64, dtype=torch.bfloat16, device=DEV)
    wantT[:C, :R] = src.t().bfloat16()
    assert torch.equal(dstT, wantT)


@pytest.mark.parametrize('M,N,K,epi', [
    (256, 3129, 512, 'tanh+mask'),        # fusion fc1 (A = 3129: ragged N, 128-column tiles)
    (256, 3129, 3129, 'bias'),            # fusion fc2: K padded to 3136, split-K
    (256, 4096, 25088, 'relu+mask'),      # fc6: one row tile, deep split-K
    (3584, 2048, 300, 'bias'),            # LSTM input projection: K = 300 -> 320
    (37, 40, 33, 'addend'),               # tiny, everything ragged (64-column tile)
    (4, 17858, 512, 'bias'),              # the token head at batch 4 x 1 token
])
def test_gemm_bf16x_vs_float64(hip_lib, bf16_mode, M, N, K, epi):
    ops = bf16_mode
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    add = torch.randn(M, N, generator=g).to(DEV) if epi == 'addend' else None
    mask = (torch.rand(M, N, generator=g) > 0.5).to(torch.uint8).to(DEV) if 'mask' in epi else None
    act = ops.ACT_TANH if 'tanh' in epi else ops.ACT_RELU if 'relu' in epi else ops.ACT_NONE
    y = ops.linear_fwd(x, W, b, act, mask, 2.0 if mask is not None else 1.0, addend=add)
    ref = r16(x) @ r16(W).t() + b.double()
    if add is not None:
        ref = ref + add.double()
    ref = torch.tanh(ref) if 'tanh' in epi else torch.relu(ref) if 'relu' in epi else ref
    if mask is not None:
        ref = ref * mask.double() * 2.0
    assert y.shape == (M, N) and torch.isfinite(y).all()
    assert rel_err(y, ref) < 2e-5
    # and the two backward products on the same operands
    dy = torch.randn(M, N, generator=g).to(DEV)
    dx = ops.linear_bwd_input(dy, W)
    assert rel_err(dx, r16(dy) @ r16(W)) < 2e-5
    dw = ops.linear_bwd_weight(dy, x)
    assert rel_err(dw, r16(dy).t() @ r16(x)) < 2e-5


def test_linear_autograd_bf16_vs_fp32(hip_lib, bf16_mode):
    """ops.linear under the bf16 dtype against the fp32 path: outputs and gradients within bf16 operand rounding."""
    ops = bf16_mode
    torch.manual_seed(4)
    x = torch.randn(64, 300, device=DEV, requires_grad=True)
    W = (torch.randn(1000, 300, device=DEV) * 0.05).requires_grad_()
    b = torch.randn(1000, device=DEV, requires_grad=True)
    gy = torch.randn(64, 1000, device=DEV)
    y16 = ops.linear(x, W, b, ops.ACT_TANH)
    g16 = torch.autograd.grad(y16, [x, W, b], gy)
    ops.set_compute_dtype('f32')
    y32 = ops.linear(x, W, b, ops.ACT_TANH)
    g32 = torch.autograd.grad(y32, [x, W, b], gy)
    ops.set_compute_dtype('bf16')
    assert rel_err(y16, y32) < 1e-2
    for a, r in zip(g16, g32):
        assert rel_err(a, r) < 1e-2


@pytest.mark.parametrize('M,K,N', [(300, 96, 1000), (37, 64, 130), (3328, 512, 17858)])
def test_fused_token_head_cross_entropy(hip_lib, bf16_mode, M, K, N):
    """LinearXentFn: loss, dX, dW, db against float64 on the bf16-rounded operands.  dl is stored as bf16 before the two backward
    products (2^-9 relative per entry): gradients are gated at 1e-2 of their largest entry, the loss at 1e-5."""
    ops = bf16_mode
    g = torch.Generator().manual_seed(M + N)
    x = torch.tanh(torch.randn(M, K, generator=g)).to(DEV).requires_grad_()
    W = (torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5).to(DEV).requires_grad_()
    b = (torch.randn(N, generator=g) * 0.1).to(DEV).requires_grad_()
    lab = torch.randint(0, N, (M,), generator=g).to(DEV)
    lab[0], lab[-1] = N - 1, 0                                    # the last valid column, next to the padding
    loss = ops.linear_cross_entropy(x, W, b, lab)
    gx, gW, gb = torch.autograd.grad(loss * 1.7, [x, W, b])
    xr, Wr, br = r16(x.detach()).requires_grad_(), r16(W.detach()).requires_grad_(), b.detach().double().requires_grad_()
    ref = torch.nn.functional.cross_entropy(xr @ Wr.t() + br, lab)
    rx, rW, rb = torch.autograd.grad(ref * 1.7, [xr, Wr, br])
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    for a, r, name in ((gx, rx, 'dx'), (gW, rW, 'dW'), (gb, rb, 'db')):
        assert a.shape == r.shape and torch.isfinite(a).all(), name
        assert rel_err(a, r) < 1e-2, (name, rel_err(a, r))
    # unfused fp32 path on the same inputs: same loss to bf16 operand rounding
    ops.set_compute_dtype('f32')
    loss32 = ops.linear_cross_entropy(x, W, b, lab)
    ops.set_compute_dtype('bf16')
    assert abs(loss.item() - loss32.item()) < 2e-3 * abs(loss32.item())


def test_w_train_step_bf16_vs_fp32(hip_lib, bf16_mode):
    """One W-model train step (fwd + CE + bwd + clip + Adam, injected dropout masks) under the bf16 dtype against the same step
    in fp32 on the small golden fixture: loss within 1e-3, every updated parameter within 2 % of the Adam step (lr)."""
    ops = bf16_mode
    from _util import full_params
    from lct_vqa_amd.basic_vqa import models, config
    from lct_vqa_amd.optim import FusedAdam
    config.DEVICE = torch.device(DEV)
    fx = load('w_small.pt')
    res = {}
    for dt in ('f32', 'bf16'):
        ops.set_compute_dtype(dt)
        m = models.VqaModel(**fx['dims'])
        m.load_state_dict(full_params(fx))
        m = m.to(DEV).train()
        opt = FusedAdam(m.trainable_parameters(), lr=1e-3, flatten=True)
        opt.zero_grad()
        with ops.inject_masks(fx['train_masks']):
            loss = ops.cross_entropy(m(fx['image'].to(DEV), fx['question'].to(DEV)), fx['label'].to(DEV))
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        opt.step(max_grad_norm=5)
        res[dt] = (loss.item(), grads, {n: p.detach().clone() for n, p in m.named_parameters()})
    ops.set_compute_dtype('bf16')
    (l32, g32, p32), (l16, g16, p16) = res['f32'], res['bf16']
    assert abs(l16 - l32) < 2e-3 * abs(l32)
    assert set(g16) == set(g32)
    for n in g32:
        assert rel_err(g16[n], g32[n]) < 5e-2, (n, rel_err(g16[n], g32[n]))


def test_ef_loss_and_gradients_bf16_vs_reference(hip_lib, bf16_mode):
    """models_lct.VqaModel._loss under the bf16 dtype (fused token head) on the small golden fixture made by the reference: loss
    1e-3, every gradient within 5e-2 of its largest entry, the same 36 tensors without a gradient."""
    ops = bf16_mode
    from _util import full_params
    from lct_vqa_amd.basic_vqa import models_lct, config
    config.DEVICE = torch.device(DEV)
    config.ARCH_TYPE = 'fixed'
    fx = load('ef_small_frozen.pt')
    m = models_lct.VqaModel(pretrained=True, **fx['dims'])
    m.load_state_dict(full_params(fx))
    m = m.to(DEV).eval()
    img, qst, lab = fx['image'].to(DEV), fx['question'].to(DEV), fx['label'].to(DEV)
    loss = m._loss(img, qst, lab)
    assert rel_err(loss, fx['loss']) < 1e-3
    grads = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
    assert sum(g is None for g in grads) == 36
    for (k, _), g in zip(m.named_parameters(), grads):
        if g is not None:
            assert rel_err(g, fx['grads'][k]) < 5e-2, (k, rel_err(g, fx['grads'][k]))


def test_resnet_gru_model_train_step_bf16(hip_lib, bf16_mode):
    """BASELINE configs[2]'s model shape under the bf16 dtype (frozen ResNet-152 trunk in bf16, GRU input projections and the
    heads through the bf16 GEMM): one train step against the same step in fp32 -- loss 2e-2 (150 layers of bf16 activations in
    the trunk), gradients of the head within 1e-1 of their largest entry, finite parameters after the update."""
    ops = bf16_mode
    from lct_vqa_amd.basic_vqa import models_ext, config
    from lct_vqa_amd.optim import FusedAdam
    config.DEVICE = torch.device(DEV)
    dims = dict(embed_size=64, qst_vocab_size=100, ans_vocab_size=80, word_embed_size=32, num_layers=1, hidden_size=64)
    g = torch.Generator().manual_seed(9)
    img = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    qst = torch.randint(0, 100, (4, 14), generator=g).to(DEV)
    lab = torch.randint(0, 80, (4,), generator=g).to(DEV)
    res = {}
    for dt in ('f32', 'bf16'):
        ops.set_compute_dtype(dt)
        torch.manual_seed(9)
        m = models_ext.VqaModelResNetGRU(pretrained=True, **dims).to(DEV).train()
        opt = FusedAdam(m.trainable_parameters(), lr=1e-3, flatten=True)
        opt.zero_grad()
        with ops.dropout_disabled():
            loss = ops.cross_entropy(m(img, qst), lab)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        opt.step(max_grad_norm=5)
        assert all(torch.isfinite(p).all() for p in m.trainable_parameters())
        res[dt] = (loss.item(), grads)
    ops.set_compute_dtype('bf16')
    (l32, g32), (l16, g16) = res['f32'], res['bf16']
    assert abs(l16 - l32) < 2e-2 * abs(l32)
    assert set(g16) == set(g32) and len(g32) > 0
    for n in g32:
        assert rel_err(g16[n], g32[n]) < 1e-1, (n, rel_err(g16[n], g32[n]))


def test_darts_vqa_weight_step_bf16_vs_fp32(hip_lib, bf16_mode):
    """BASELINE configs[4]'s model (darts_vqa VqaModel with the PC-DARTS search encoder) under the bf16 compute dtype at its own
    shapes (224x224, QV = 17 858, A = 3129, B = 2): the heads -- encoder fc 12544 -> 512, LSTM input projection, fused token head +
    cross entropy, fusion fc1 / fc2 -- run on the bf16 GEMMs; the MixedOp edges compute in fp32 and STORE their nine pre-BatchNorm branch
    tensors in bf16 (round 4; pinned kernel by kernel against the oracle with the same roundings at 1e-3:
    tests/test_gpu_edge.py::test_fused_edge_bf16_storage_vs_oracle).  One weight step against the same step in fp32: loss 2e-3; the
    gradients of the heads and of alpha / beta within 5e-2 of their largest entry; the encoder's parameters -- whose gradients cross up to
    56 edges in which a pre-activation within one bf16 rounding of zero takes the other ReLU branch than in fp32 -- cosine >= 0.97 and
    the largest deviation of an entry within 3.5e-1 of the tensor's largest entry (measured at this B = 2: cosine >= 0.994, 0.23 on one
    16 x 16 pointwise weight, 0.09 at the stem).  The fused token-head path is really taken (no [B,T,QV] logits tensor)."""
    ops = bf16_mode
    from lct_vqa_amd.darts_vqa import config, vqa_model
    from lct_vqa_amd.basic_vqa import config as bconfig
    config.DEVICE = bconfig.DEVICE = torch.device(DEV)
    dims = dict(embed_size=512, qst_vocab_size=17858, ans_vocab_size=3129, word_embed_size=300, num_layers=1, hidden_size=512)
    g = torch.Generator().manual_seed(21)
    img = torch.randn(2, 3, 224, 224, generator=g).to(DEV)
    qst = torch.randint(0, dims['qst_vocab_size'], (2, 14), generator=g).to(DEV)
    lab = torch.randint(0, dims['ans_vocab_size'], (2,), generator=g).to(DEV)
    torch.manual_seed(21)
    m = vqa_model.VqaModel(img_encoder_type='darts', **dims).to(DEV).train()
    params = list(m.parameters()) + list(m.arch_parameters())
    res = {}
    for dt in ('f32', 'bf16'):
        ops.set_compute_dtype(dt)
        state = {k: v.clone() for k, v in m.state_dict().items()}          # BatchNorm running statistics move in train mode
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        with ops.dropout_disabled():
            loss = m._loss(img, qst, lab)
        peak = torch.cuda.max_memory_allocated() - base
        gr = torch.autograd.grad(loss, params)
        m.load_state_dict(state)
        res[dt] = (loss.item(), gr, peak)
    ops.set_compute_dtype('bf16')
    (l32, g32, p32), (l16, g16, p16) = res['f32'], res['bf16']
    assert abs(l16 - l32) < 2e-3 * abs(l32)
    for (n, _), a, r in zip(list(m.named_parameters()) + [(f'arch{i}', None) for i in range(4)], g16, g32):
        assert torch.isfinite(a).all(), n
        if n.startswith('img_encoder.darts'):
            cos = torch.nn.functional.cosine_similarity(a.flatten().double(), r.flatten().double(), dim=0).item()
            assert rel_err(a, r) < 3.5e-1 and cos > 0.97, (n, rel_err(a, r), cos)
        else:
            assert rel_err(a, r) < 5e-2, (n, rel_err(a, r))


def test_trainable_resnet_gru_bf16_vs_fp32(hip_lib, bf16_mode):
    """BASELINE configs[2] with the encoder TRAINED ('encoder trainable') under the bf16 dtype: the ResNet-152's 1x1
    whole trunk runs bf16-native (resnet_bf16.py), the heads on the bf16 GEMM.  Forward + backward of the full-depth model against the
    same in fp32, with the BatchNorms on (randomised) running statistics: logits 5e-2, gradients of the heads within 1.5e-1 of their
    largest entry, of trunk parameters from every stage cosine >= 0.8.  (Batch-statistics BatchNorm at a size where it means something --
    224^2, B = 8 -- is tests/test_gpu_train_bf16.py::test_trainable_trunk_bf16_batch_stat_bn_vs_oracles.)"""
    ops = bf16_mode
    from test_gpu_resnet import _randomise_bn
    from lct_vqa_amd.basic_vqa import models_ext, config
    config.DEVICE = torch.device(DEV)
    dims = dict(embed_size=64, qst_vocab_size=100, ans_vocab_size=80, word_embed_size=32, num_layers=1, hidden_size=64)
    g = torch.Generator().manual_seed(19)
    img = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    qst = torch.randint(0, 100, (4, 14), generator=g).to(DEV)
    lab = torch.randint(0, 80, (4,), generator=g).to(DEV)
    names = ['img_encoder.model.conv1.weight', 'img_encoder.model.layer1.0.conv1.weight', 'img_encoder.model.layer2.3.conv3.weight',
             'img_encoder.model.layer3.20.conv2.weight', 'img_encoder.model.layer4.2.bn3.weight', 'img_encoder.fc.weight', 'fc2.weight']
    torch.manual_seed(19)
    m = models_ext.VqaModelResNetGRU(pretrained=False, **dims)
    _randomise_bn(m.img_encoder.model, 20)
    m = m.to(DEV).eval()                                          # BatchNorm on running statistics; gradients still wanted: trainable path
    pd = dict(m.named_parameters())
    res = {}
    for dt in ('f32', 'bf16'):
        ops.set_compute_dtype(dt)
        logits = m(img, qst)
        assert logits.grad_fn is not None
        loss = ops.cross_entropy(logits, lab)
        grads = torch.autograd.grad(loss, [pd[n] for n in names])
        res[dt] = (logits.detach(), loss.item(), grads)
    ops.set_compute_dtype('bf16')
    (y32, l32, g32), (y16, l16, g16) = res['f32'], res['bf16']
    assert rel_err(y16, y32) < 5e-2 and abs(l16 - l32) < 1e-2 * abs(l32)
    for n, a, r in zip(names, g16, g32):
        assert torch.isfinite(a).all(), n
        # heads: 1.5e-1 of the largest entry.  Inside the trunk the bf16-native path (round 4: bf16 activations end to end,
        # resnet_bf16.py) is held to the gradient's direction, cosine >= 0.8 through up to 150 layers: bf16 storage flips the ReLU of
        # pre-activations within one rounding of zero and each flip moves single entries by their own size (the tight pins of this
        # path are tests/test_gpu_train_bf16.py: block level 1e-2 against the bf16-storage emulation of the fp32 oracle)
        if '.model.' in n:
            cos = (a.double().flatten() @ r.double().flatten() / (a.double().norm() * r.double().norm())).item()
            assert cos > 0.8, (n, cos)
        else:
            assert rel_err(a, r) < 1.5e-1, (n, rel_err(a, r))
