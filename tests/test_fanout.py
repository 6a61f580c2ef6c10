"""darts_ops.FanoutFn / GradSlot on the CPU: the protocol that lets the readers of one state accumulate their input gradients into a
single buffer (the fused MixedOp edges do it inside their kernels).  Toy readers stand in for the edges."""
import torch

from lct_vqa_amd import darts_ops as D


class _Reader(torch.autograd.Function):
    """y = k * x; with a slot its input gradient goes into the slot's buffer and None is returned (what MixedEdgeFn does)."""

    @staticmethod
    def forward(ctx, x, k, slot):
        ctx.k, ctx.slot = k, slot
        return x * k

    @staticmethod
    def backward(ctx, g):
        if ctx.slot is None:
            return g * ctx.k, None, None
        buf, acc = ctx.slot.target(g)
        if acc:
            buf.add_(g * ctx.k)
        else:
            buf.copy_(g * ctx.k)
        return None, None, None


def test_fanout_sums_slot_and_ordinary_gradients_over_repeated_passes():
    w = torch.ones(5, requires_grad=True)
    slot = D.GradSlot()
    for _ in range(3):                                              # the slot is reused by every pass; each pass gets a fresh buffer
        a, b, c, d = D.FanoutFn.apply(w * 1.0, 4, slot)
        y = _Reader.apply(a, 2.0, slot) + _Reader.apply(b, 3.0, slot) + _Reader.apply(c, 5.0, None) + d * 7.0
        g, = torch.autograd.grad(y.sum(), w)
        assert torch.equal(g, torch.full((5,), 17.0))
        assert slot.buf is None                                     # handed on, not kept


def test_fanout_when_only_some_readers_are_reached():
    """A reader whose output does not reach the loss never runs; FanoutFn.backward still returns what the others left."""
    w = torch.ones(3, requires_grad=True)
    slot = D.GradSlot()
    a, b = D.FanoutFn.apply(w * 1.0, 2, slot)
    used, _unused = _Reader.apply(a, 2.0, slot), _Reader.apply(b, 3.0, slot)
    g, = torch.autograd.grad(used.sum(), w)
    assert torch.equal(g, torch.full((3,), 2.0))


def test_fanout_under_no_grad_returns_plain_aliases():
    x = torch.arange(4.0)
    with torch.no_grad():
        a, b = D.FanoutFn.apply(x, 2, D.GradSlot())
    assert a.data_ptr() == x.data_ptr() == b.data_ptr() and not a.requires_grad
