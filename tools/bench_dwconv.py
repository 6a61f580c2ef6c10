#!/usr/bin/env python3
"""Time the depthwise convolution kernels (forward, input gradient, weight gradient; ReLU in front fused) at the PC-DARTS
search network's shapes (B=16, 224x224 input) and check them against torch on the same data.
    python tools/bench_dwconv.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd._lib import lib  # noqa: E402

#         B   H   C  K  s  p  d
SHAPES = [(16, 224, 4, 5, 1, 2, 1), (16, 224, 4, 5, 1, 4, 2), (16, 224, 4, 3, 1, 1, 1), (16, 112, 8, 5, 1, 2, 1), (16, 112, 8, 3, 1, 1, 1),
          (16, 56, 16, 5, 1, 2, 1), (16, 56, 16, 3, 1, 2, 2), (16, 224, 4, 5, 2, 2, 1), (16, 112, 8, 3, 2, 1, 1)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def main():
    dev = torch.device('cuda')
    for (B, H, C, K, s, p, d) in SHAPES:
        Ho = (H + 2 * p - d * (K - 1) - 1) // s + 1
        x = torch.randn(B, H, H, C, device=dev)
        w = torch.randn(C, K, K, device=dev)
        dy = torch.randn(B, Ho, Ho, C, device=dev)
        y, dx, dw = torch.empty(B, Ho, Ho, C, device=dev), torch.empty_like(x), torch.empty(C, K, K, device=dev)
        scratch = torch.empty(lib.query('dwconv_bwd_weight_scratch_elems', B, H, H, C, K, s, p, d), device=dev)
        t_f = timed(lambda: lib('dwconv_fwd', x, w, y, B, H, H, C, K, s, p, d, 1))
        t_i = timed(lambda: lib('dwconv_bwd_input', dy, w, x, dx, B, H, H, C, K, s, p, d, 1))
        t_w = timed(lambda: lib('dwconv_bwd_weight', dy, x, dw, scratch, B, H, H, C, K, s, p, d, 1))
        # torch on the same data (NCHW views)
        xr = x.permute(0, 3, 1, 2).detach().clone().requires_grad_()
        wr = w.view(C, 1, K, K).detach().clone().requires_grad_()
        yr = F.conv2d(torch.relu(xr), wr, stride=s, padding=p, dilation=d, groups=C)
        gx, gw = torch.autograd.grad(yr, [xr, wr], dy.permute(0, 3, 1, 2))
        e = (rel(y.permute(0, 3, 1, 2), yr), rel(dx.permute(0, 3, 1, 2), gx), rel(dw, gw[:, 0]))
        gb = x.numel() * 4 / 1e9
        print(f'B{B} {H}x{H} C{C} K{K} s{s} d{d}: fwd {t_f:6.1f} us  dgrad {t_i:6.1f} us  wgrad {t_w:6.1f} us   '
              f'(x = {gb * 1e3:.1f} MB)  rel err {e[0]:.1e} {e[1]:.1e} {e[2]:.1e}')


if __name__ == '__main__':
    main()
