"""CPU tests of the shared architecture-gradient machinery (lct_vqa_amd/bilevel.py is pure tensor algebra): the pieces against
closed forms / autograd's exact second derivatives on a toy bilevel problem, and a whole 2-level DARTS step against the exact
hypergradient."""
import pytest
import torch

from lct_vqa_amd import bilevel


class Toy(torch.nn.Module):
    """L(w, a) = mean((x @ (w * softplus(a)) - y)^2): weights w (two tensors), architecture parameter a."""

    def __init__(self, d=6):
        super().__init__()
        f64 = dict(dtype=torch.float64)
        self.w1 = torch.nn.Parameter(torch.randn(d, **f64))
        self.w2 = torch.nn.Parameter(torch.randn(d, **f64))
        self.unused = torch.nn.Parameter(torch.randn(3, **f64))
        self.register_buffer('count', torch.zeros((), **f64))
        self.a = (0.1 * torch.randn(d, **f64)).requires_grad_(True)

    def arch_parameters(self):
        return [self.a]

    def new(self):
        twin = Toy(self.w1.numel())
        twin.a.data.copy_(self.a.data)
        return twin

    def _loss(self, x, y, _label=None):
        gate = torch.nn.functional.softplus(self.a)
        return ((x @ (self.w1 * gate) + (x ** 2) @ (self.w2 * gate) - y) ** 2).mean()


def _data(seed, n=32, d=6):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g, dtype=torch.float64), torch.randn(n, generator=g, dtype=torch.float64)


def test_grads_zero_fill_and_unreached_count():
    torch.manual_seed(0)
    m = Toy()
    x, y = _data(1)
    g = bilevel.grads(m._loss(x, y), m.parameters(), expect_unused=1)
    assert [t.shape for t in g] == [p.shape for p in m.parameters()]
    assert torch.count_nonzero(g[2]) == 0 and torch.count_nonzero(g[0]) > 0
    with pytest.raises(AssertionError):
        bilevel.grads(m._loss(x, y), m.parameters(), expect_unused=0)
    assert abs(bilevel.global_norm(g).item() - torch.cat([t.reshape(-1) for t in g]).norm().item()) < 1e-12


def test_virtual_step_is_an_independent_sgd_step():
    torch.manual_seed(1)
    m = Toy()
    m.count.fill_(7.0)
    x, y = _data(2)
    before = [p.detach().clone() for p in m.parameters()]
    g = bilevel.grads(m._loss(x, y), m.parameters(), 1)
    twin = bilevel.virtual_step(m, g, 0.05)
    for p, q, b, gi in zip(m.parameters(), twin.parameters(), before, g):
        assert torch.equal(p.detach(), b)                                  # the original is untouched
        assert torch.allclose(q.detach(), b - 0.05 * gi, rtol=0, atol=1e-15)
        assert q.data_ptr() != p.data_ptr()
    assert twin.count.item() == 7.0 and torch.equal(twin.a.detach(), m.a.detach()) and twin.a is not m.a


def test_fd_grad_matches_exact_mixed_second_derivative_and_restores_parameters():
    torch.manual_seed(2)
    m = Toy()
    x, y = _data(3)
    weights = [m.w1, m.w2]
    v = [torch.randn_like(p) for p in weights]
    before = [p.detach().clone() for p in weights]
    fd = bilevel.fd_grad(weights, v, lambda: bilevel.grads(m._loss(x, y), m.arch_parameters()), r=1e-3)
    gw = torch.autograd.grad(m._loss(x, y), weights, create_graph=True)
    exact = torch.autograd.grad(sum((a * b).sum() for a, b in zip(gw, v)), m.a)[0]     # d/da (grad_w L . v)
    assert torch.allclose(fd[0], exact, rtol=1e-5, atol=1e-8)
    for p, b in zip(weights, before):
        assert torch.allclose(p.detach(), b, rtol=0, atol=1e-14)


def test_two_level_architect_step_matches_exact_hypergradient(monkeypatch):
    """Architect.compute_grads on the toy problem = d L_val(w - eta grad_w L_train(w, a), a) / da by exact autograd."""
    from lct_vqa_amd.basic_vqa import config
    from lct_vqa_amd.basic_vqa.pcdarts.architect import Architect
    monkeypatch.setattr(config, 'DEVICE', torch.device('cpu'))
    torch.manual_seed(3)
    m = Toy()
    xt, yt = _data(4)
    xv, yv = _data(5)
    eta = 0.05
    arch = Architect(m)
    arch.compute_grads(xt, yt, None, xv, yv, None, eta=eta)
    # exact: differentiate through the virtual step
    gw = torch.autograd.grad(m._loss(xt, yt), [m.w1, m.w2], create_graph=True)
    w1, w2 = m.w1 - eta * gw[0], m.w2 - eta * gw[1]
    gate = torch.nn.functional.softplus(m.a)
    val = ((xv @ (w1 * gate) + (xv ** 2) @ (w2 * gate) - yv) ** 2).mean()
    exact = torch.autograd.grad(val, m.a)[0]
    assert torch.allclose(m.a.grad, exact, rtol=2e-4, atol=1e-7)
    a_before = m.a.detach().clone()
    arch.optimizer.step()
    assert not torch.equal(m.a.detach(), a_before)


def test_cached_twin_is_indistinguishable_from_a_fresh_one():
    """virtual_step(..., cache): the twin of the previous step is re-used; after the model's parameters, buffers and
    architecture parameters moved it must hold exactly what a fresh new() + load_state_dict + SGD step would."""
    torch.manual_seed(3)
    m = Toy()
    x, y = _data(4)
    cache = {}
    first = bilevel.virtual_step(m, bilevel.grads(m._loss(x, y), m.parameters(), 1), 0.05, cache)
    assert cache['twin'] is first
    with torch.no_grad():                                       # the model moves on: weights, buffer, architecture parameter
        for p in m.parameters():
            p.add_(0.3 * torch.randn_like(p))
        m.count.fill_(11.0)
        m.a.add_(0.2)
    g = bilevel.grads(m._loss(x, y), m.parameters(), 1)
    again = bilevel.virtual_step(m, g, 0.05, cache)
    fresh = bilevel.virtual_step(m, g, 0.05)
    assert again is first and fresh is not first
    for (k, a), (_, b) in zip(again.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(again.a, fresh.a) and again.a.data_ptr() != m.a.data_ptr()
    other = Toy(4)                                              # a different model under the same cache: starts over
    t = bilevel.virtual_step(other, [torch.zeros_like(p) for p in other.parameters()], 0.1, cache)
    assert t is not first and t.w1.shape == (4,)


def test_architecture_passes_keep_fp32_edge_storage_and_restore_the_setting():
    """Every architect's compute_grads is wrapped in bilevel.architecture_pass: inside, the fused MixedOp edges store their pre-BN tensors in
    fp32 whatever the compute dtype (a finite difference of two gradients does not survive an 8-bit mantissa); the caller's setting comes
    back afterwards, also when the pass raises."""
    from lct_vqa_amd import darts_ops, ops
    from lct_vqa_amd.basic_vqa.pcdarts import architect, architect_lct
    from lct_vqa_amd.darts_vqa.pcdarts import architect_vqa
    for cls in (architect.Architect, architect_lct.ArchitectLct, architect_vqa.Architect):
        assert getattr(cls.compute_grads, '__wrapped__', None) is not None, cls
    prev_dtype = ops.set_compute_dtype('bf16')
    prev = darts_ops.set_edge_pre_bf16(None)
    try:
        assert darts_ops.edge_pre_bf16() is True                  # follows the compute dtype
        seen = []

        @bilevel.architecture_pass
        def inside(fail):
            seen.append(darts_ops.edge_pre_bf16())
            if fail:
                raise RuntimeError('boom')
            return 7

        assert inside(False) == 7 and seen == [False]
        assert darts_ops.edge_pre_bf16() is True
        with pytest.raises(RuntimeError):
            inside(True)
        assert darts_ops.edge_pre_bf16() is True and darts_ops.EDGE_PRE_BF16 is None
        darts_ops.set_edge_pre_bf16(False)
        assert darts_ops.edge_pre_bf16() is False
    finally:
        darts_ops.set_edge_pre_bf16(prev)
        ops.set_compute_dtype(prev_dtype)
