#!/usr/bin/env python3
"""Fused MixedOp edge kernels (csrc/mixed_edge.hip) against the per-primitive kernels (which are pinned to the reference):
output, running statistics and EVERY gradient, for all channel widths / strides, train and eval mode, with and without a
running node sum.  Prints the worst relative deviation per case."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(C, stride, B, H, W, training, with_prev, seed=0):
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import MixedOp
    torch.manual_seed(seed)
    op = MixedOp(C, stride).cuda()
    for m in op.modules():                                     # non-trivial running statistics for the eval case
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
    op.train(training)
    x0 = torch.randn(B, H, W, C, device='cuda')
    w0 = torch.softmax(torch.randn(8, device='cuda'), 0)
    b0 = torch.rand(1, device='cuda') + 0.5
    Ho, Wo = H // stride, W // stride
    prev0 = torch.randn(B, Ho, Wo, C, device='cuda') if with_prev else None
    gout = torch.randn(B, Ho, Wo, C, device='cuda')
    state0 = {k: v.clone() for k, v in op.state_dict().items()}
    res = {}
    for fused in (False, True):
        D.set_fused_edge(fused)
        op.load_state_dict(state0)
        x, w, b = x0.clone().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
        prev = (prev0 * 1.0).requires_grad_() if with_prev else None
        node = prev * 1.0 if with_prev else None               # non-leaf: the op updates it in place
        out = op(x, w, node, b)
        params = [p for p in op.parameters()]
        grads = torch.autograd.grad(out, [x, w, b] + ([prev] if with_prev else []) + params, gout, allow_unused=True)
        res[fused] = (out.detach().clone(), grads, {k: v.clone() for k, v in op.state_dict().items()}, [n for n, _ in op.named_parameters()])
    (o0, g0, s0, names), (o1, g1, s1, _) = res[False], res[True]
    worst = {'out': rel(o1, o0)}
    labels = ['dx', 'dw', 'dbeta'] + (['dprev'] if with_prev else []) + names
    for lab, a, b_ in zip(labels, g1, g0):
        if (a is None) != (b_ is None):
            worst[lab] = float('inf')
        elif a is not None:
            worst[lab] = rel(a, b_)
    for k in s0:
        if s0[k].is_floating_point():
            worst['buf ' + k] = rel(s1[k], s0[k])
        elif not torch.equal(s0[k], s1[k]):
            worst['buf ' + k] = float('inf')
    bad = {k: v for k, v in worst.items() if not v < 2e-4}
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f'C={C:3d} s={stride} B={B} {H}x{W} train={int(training)} prev={int(with_prev)}: worst {top}' + (f'  BAD {bad}' if bad else ''), flush=True)
    return not bad


def main():
    from lct_vqa_amd._lib import lib
    from lct_vqa_amd.basic_vqa import config
    lib.load()
    config.DEVICE = torch.device('cuda')
    ok = True
    cases = [(16, 1, 3, 24, 40), (16, 2, 2, 24, 40), (32, 1, 2, 16, 36), (32, 2, 3, 20, 68), (64, 1, 2, 14, 14), (64, 2, 2, 28, 36),
             (16, 1, 2, 8, 8), (64, 2, 1, 8, 8)]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for C, s, B, H, W in cases:
        for training, with_prev in ((True, False), (True, True), (False, True)):
            ok &= run(C, s, B, H, W, training, with_prev)
    print('ALL OK' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
