"""GPU parity of the PC-DARTS primitives (vs plain torch fp32 CPU ops = the arithmetic the reference's modules
dispatch) and of the whole search network vs the golden vectors produced by the reference's own Network."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from _util import load, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.fixture(scope='module')
def D(hip_lib):
    from lct_vqa_amd import darts_ops
    return darts_ops


@pytest.mark.parametrize('K,stride,pad,dil', [(3, 1, 1, 1), (5, 1, 2, 1), (3, 2, 1, 1), (5, 2, 2, 1), (3, 1, 2, 2), (5, 1, 4, 2), (3, 2, 2, 2), (5, 2, 4, 2)])
def test_dwconv(D, K, stride, pad, dil):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 12, 10, generator=g, requires_grad=True)
    w = torch.randn(8, 1, K, K, generator=g, requires_grad=True)
    ref = F.conv2d(torch.relu(x), w, None, stride, pad, dil, 8)
    gy = torch.randn_like(ref)
    gx, gw = torch.autograd.grad(ref, [x, w], gy)
    xd, wd = nhwc(x.detach()).requires_grad_(), w.detach().to(DEV).requires_grad_()
    y = D.DwConvFn.apply(xd, wd, stride, pad, dil, True)
    assert rel_err(nchw(y), ref) < 1e-5
    dx, dw = torch.autograd.grad(y, [xd, wd], nhwc(gy))
    assert rel_err(nchw(dx), gx) < 1e-5 and rel_err(dw, gw) < 1e-4


@pytest.mark.parametrize('Cout,Cin', [(16, 48), (32, 48), (32, 64), (64, 128), (64, 256)])
@pytest.mark.parametrize('relu', [True, False])
def test_preprocess_pointwise_conv_shapes_vs_torch(D, Cout, Cin, relu):
    """ReLUConvBN's 1x1 convolution at the channel counts of the search network's preprocess layers (operations.py:24-33: 3C = 48 stem
    channels and 4C cell outputs into C): forward, input gradient (with the ReLU mask) and weight gradient, each on the kernel the
    dispatch picks for that shape (GEMM / pwconv_small / pwconv_wgrad_blk)."""
    g = torch.Generator().manual_seed(Cout + Cin)
    x = torch.randn(2, Cin, 13, 11, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).requires_grad_()
    ref = F.conv2d(torch.relu(x) if relu else x, w)
    gy = torch.randn_like(ref)
    gx, gw = torch.autograd.grad(ref, [x, w], gy)
    xd, wd = nhwc(x.detach()).requires_grad_(), w.detach().to(DEV).requires_grad_()
    y = D.PwConvFn.apply(xd, wd, relu)
    assert rel_err(nchw(y), ref) < 1e-5
    dx, dw = torch.autograd.grad(y, [xd, wd], nhwc(gy))
    assert rel_err(nchw(dx), gx) < 1e-5 and rel_err(dw, gw) < 1e-5


def test_pwconv_and_bn(D):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 12, 9, 7, generator=g, requires_grad=True)
    w = torch.randn(20, 12, 1, 1, generator=g, requires_grad=True)
    ref = F.conv2d(torch.relu(x), w)
    gy = torch.randn_like(ref)
    gx, gw = torch.autograd.grad(ref, [x, w], gy)
    xd, wd = nhwc(x.detach()).requires_grad_(), w.detach().to(DEV).requires_grad_()
    y = D.PwConvFn.apply(xd, wd, True)
    assert rel_err(nchw(y), ref) < 1e-5
    dx, dw = torch.autograd.grad(y, [xd, wd], nhwc(gy))
    assert rel_err(nchw(dx), gx) < 1e-5 and rel_err(dw, gw) < 1e-4
    for affine in (False, True):
        for train in (True, False):
            bn = nn.BatchNorm2d(12, affine=affine)
            bn.running_mean.normal_(generator=g); bn.running_var.uniform_(0.5, 2.0, generator=g)
            if affine:
                bn.weight.data.normal_(generator=g); bn.bias.data.normal_(generator=g)
            bnd = nn.BatchNorm2d(12, affine=affine)
            bnd.load_state_dict(bn.state_dict())
            bnd = bnd.to(DEV)
            bn.train(train); bnd.train(train)
            xr = x.detach().clone().requires_grad_()
            yr = bn(xr)
            gy = torch.randn_like(yr)
            ps = [xr] + (list(bn.parameters()) if affine else [])
            gr = torch.autograd.grad(yr, ps, gy)
            xd = nhwc(x.detach()).requires_grad_()
            yd = D.batchnorm(bnd, xd)
            assert rel_err(nchw(yd), yr) < 1e-5
            gd = torch.autograd.grad(yd, [xd] + (list(bnd.parameters()) if affine else []), nhwc(gy))
            assert rel_err(nchw(gd[0]), gr[0]) < 1e-4
            for a, r in zip(gd[1:], gr[1:]):
                assert rel_err(a, r) < 1e-4
            assert rel_err(bnd.running_mean, bn.running_mean) < 1e-5 and rel_err(bnd.running_var, bn.running_var) < 1e-5
            assert int(bnd.num_batches_tracked) == int(bn.num_batches_tracked)


@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('kind', [0, 1])
def test_pool3x3(D, stride, kind):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 10, 12, generator=g, requires_grad=True)
    ref = F.max_pool2d(x, 3, stride, 1) if kind == 0 else F.avg_pool2d(x, 3, stride, 1, count_include_pad=False)
    gy = torch.randn_like(ref)
    (gx,) = torch.autograd.grad(ref, [x], gy)
    xd = nhwc(x.detach()).requires_grad_()
    y = D.Pool3x3Fn.apply(xd, stride, kind)
    assert rel_err(nchw(y), ref) < 1e-6
    (dx,) = torch.autograd.grad(y, [xd], nhwc(gy))
    assert rel_err(nchw(dx), gx) < 1e-5


def test_factorized_reduce_and_mixed_op(D, hip_lib):
    from lct_vqa_amd.basic_vqa import config
    config.DEVICE = torch.device(DEV)
    from lct_vqa_amd.basic_vqa.pcdarts import model_search as MS
    from oracle import darts_oracle as DO
    torch.manual_seed(4)
    for stride in (1, 2):
        mo = MS.MixedOp(16, stride)
        P = {'m.' + k: v.detach().clone() for k, v in mo.state_dict().items()}
        x = torch.randn(2, 16, 8, 8)
        w = torch.softmax(torch.randn(8), 0)
        beta = torch.tensor([0.7])
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), beta.clone().requires_grad_()
        ns = {}
        ref = br * DO._mixed_op(xr, wr, P, 'm', stride, True, ns)
        gy = torch.randn_like(ref)
        gx, gw, gb = torch.autograd.grad(ref, [xr, wr, br], gy)
        mod = mo.to(DEV).train()
        xd, wd, bd = nhwc(x).requires_grad_(), w.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
        y = mod(xd, wd, None, bd)
        assert rel_err(nchw(y), ref) < 1e-4
        dx, dw, db = torch.autograd.grad(y, [xd, wd, bd], nhwc(gy))
        assert rel_err(nchw(dx), gx) < 1e-3 and rel_err(dw, gw) < 1e-3 and rel_err(db, gb) < 1e-3
        # accumulate into an existing state (second edge of a node)
        y2 = mod(xd, wd, y.detach().clone(), bd)
        assert rel_err(nchw(y2), 2 * ref) < 1e-4


def test_network_against_reference_goldens(D):
    from lct_vqa_amd.basic_vqa import config
    config.DEVICE = torch.device(DEV)
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
    fx = load('darts_net.pt')
    torch.manual_seed(fx['seed'])
    net = Network(fx['C'], 512, fx['layers'])
    # same seed + same construction order => identical initial parameters as the reference Network
    sd = net.state_dict()
    assert set(sd) == set(fx['state'])
    for k, v in fx['state'].items():
        assert torch.equal(sd[k].cpu(), v), k
    net = net.to(DEV)
    for a, r in zip(net.arch_parameters(), fx['arch']):
        a.data.copy_(r)
    assert str(net.genotype()) == fx['genotype']
    img = fx['image'].to(DEV)
    net.eval()
    with torch.no_grad():
        ev = net(img)
    assert rel_err(ev, fx['eval_out']) < 1e-3
    net.train()
    tr = net(img)
    assert rel_err(tr, fx['train_out']) < 1e-3
    names = list(fx['param_grads'])
    pd = dict(net.named_parameters())
    grads = torch.autograd.grad((tr * fx['loss_weight'].to(DEV)).sum(), list(net.arch_parameters()) + [pd[n] for n in names])
    for a, r in zip(grads[:4], fx['arch_grads']):
        assert rel_err(a, r) < 2e-3
    for n, a in zip(names, grads[4:]):
        assert rel_err(a, fx['param_grads'][n]) < 2e-3, n
    sd = net.state_dict()
    for k, v in fx['running_after'].items():
        assert rel_err(sd[k], v) < 1e-4, k


def test_derived_network_from_searched_genotype_vs_oracle(D):
    """SURVEY f-4: decode the searched architecture (the golden fixture's alphas/betas -> the reference's genotype string)
    and run the DERIVED network (one op per edge, two edges per node) on the HIP primitives: forward in eval and train
    mode, every parameter gradient and the BatchNorm running statistics against the oracle restatement.  A second,
    hand-written genotype covers every primitive (both pools, skip/FactorizedReduce, sep 3/5, dil 3/5) at stride 1 and 2."""
    from oracle import darts_oracle as O
    from lct_vqa_amd.basic_vqa import config
    config.DEVICE = torch.device(DEV)
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
    from lct_vqa_amd.basic_vqa.pcdarts.model import NetworkDerived
    from lct_vqa_amd.basic_vqa.pcdarts.genotypes import Genotype
    fx = load('darts_net.pt')
    torch.manual_seed(fx['seed'])
    search = Network(fx['C'], 512, fx['layers']).to(DEV)
    for a, r in zip(search.arch_parameters(), fx['arch']):
        a.data.copy_(r)
    searched = search.genotype()
    assert str(searched) == fx['genotype']
    every = Genotype(
        normal=[('sep_conv_3x3', 0), ('max_pool_3x3', 1), ('skip_connect', 0), ('dil_conv_5x5', 2),
                ('avg_pool_3x3', 1), ('sep_conv_5x5', 3), ('dil_conv_3x3', 2), ('skip_connect', 4)],
        normal_concat=range(2, 6),
        reduce=[('max_pool_3x3', 0), ('skip_connect', 1), ('sep_conv_5x5', 0), ('avg_pool_3x3', 2),
                ('dil_conv_3x3', 1), ('dil_conv_5x5', 0), ('sep_conv_3x3', 3), ('avg_pool_3x3', 1)],
        reduce_concat=range(2, 6))
    g = torch.Generator().manual_seed(5)
    img = torch.randn(3, 3, 32, 32, generator=g)
    for geno in (searched, every):
        torch.manual_seed(21)
        net = NetworkDerived(fx['C'], 512, fx['layers'], geno)
        for n, b in net.named_buffers():                       # non-trivial running statistics for the eval pass
            if n.endswith('running_mean'):
                b.normal_(0, 0.1, generator=g)
            elif n.endswith('running_var'):
                b.uniform_(0.5, 1.5, generator=g)
        P = {k: v.clone() for k, v in net.state_dict().items()}
        net = net.to(DEV)
        net.eval()
        with torch.no_grad():
            ev = net(img.to(DEV))
        ref_ev = O.derived_forward(img, P, geno, train=False, layers=fx['layers'])
        assert ev.shape == ref_ev.shape == (3, net.output_ch * 49)
        assert rel_err(ev, ref_ev) < 1e-4
        net.train()
        tr = net(img.to(DEV))
        Pg = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running' not in k else v.clone()) for k, v in P.items()}
        ns = {}
        ref_tr = O.derived_forward(img, Pg, geno, train=True, new_stats=ns, layers=fx['layers'])
        assert rel_err(tr, ref_tr) < 1e-4
        wgt = torch.randn(ref_tr.shape, generator=g)
        names = [n for n, _ in net.named_parameters()]
        ref_g = torch.autograd.grad((ref_tr * wgt).sum(), [Pg[n] for n in names])
        got_g = torch.autograd.grad((tr * wgt.to(DEV)).sum(), [p for _, p in net.named_parameters()])
        for n, a, r in zip(names, got_g, ref_g):
            assert rel_err(a, r) < 2e-3, n
        sd = net.state_dict()
        for k, v in ns.items():
            assert rel_err(sd[k], v) < 1e-4, k


def test_search_network_step_as_hip_graph(hip_lib):
    """The C ABI only launches on the stream it is given, so a whole forward+backward of the search network can be captured
    into one HIP graph (lct_vqa_amd.graph.Graphed) and replayed: same gradients as the eager step, bit for bit, also after
    the parameters and the input have been changed in place (the graph reads them at replay time)."""
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
    from lct_vqa_amd.graph import Graphed
    config.DEVICE = torch.device(DEV)
    torch.manual_seed(3)
    net = Network(16, 512, 4).to(DEV).train()
    x = torch.randn(2, 3, 32, 32, device=DEV)
    params = list(net.parameters()) + net.arch_parameters()
    step = lambda: torch.autograd.grad(net(x).sum(), params)
    g = Graphed(step)
    for trial in range(2):
        got = [t.clone() for t in g.replay()]
        ref = step()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        with torch.no_grad():                                   # move parameters and input in place, replay again
            for p_ in params:
                p_.add_(0.01 * torch.randn_like(p_))
            x.copy_(torch.randn_like(x))


def test_graphed_refuses_a_step_that_would_crash_the_capture(hip_lib):
    """The capture crash on record (gpurun_out/graph_fault.log) was preceded by torch's AccumulateGrad stream-mismatch
    warning: a ``backward()`` into ``.grad`` whose AccumulateGrad nodes belong to an older graph on the default stream.
    ``Graphed`` must see that during its warm-up and raise instead of starting a capture; the same step is fine once the old
    graph is gone, and a ``backward()`` step whose nodes are made during warm-up captures and replays correctly."""
    from lct_vqa_amd.graph import Graphed, GraphCaptureError
    w = torch.randn(64, 64, device=DEV, requires_grad=True)
    x = torch.randn(8, 64, device=DEV)
    old_loss = (x @ w).sum()                       # an eager iteration whose graph (and AccumulateGrad node of w) stays alive
    old_loss.backward(retain_graph=True)

    def step():
        (x @ w).pow(2).sum().backward()
        return w.grad

    with pytest.raises(GraphCaptureError):
        Graphed(step)
    del old_loss
    w.grad = None
    g = Graphed(step)
    w.grad.zero_()
    out = g.replay()
    w2 = w.detach().clone().requires_grad_()
    (x @ w2).pow(2).sum().backward()
    assert torch.allclose(out, w2.grad, rtol=1e-5, atol=1e-5)
