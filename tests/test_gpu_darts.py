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
    from lct_vqa_amd.basic_vqa.pcdarts import operations as OP, model_search as MS
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
