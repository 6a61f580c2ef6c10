#!/usr/bin/env python3
"""Asymptote of the LDS-DMA fp32 GEMM: one large product and the 36 transformed-domain products of a Winograd layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd._lib import lib  # noqa: E402

dev = torch.device('cuda')


def timeit(fn, n=5):
    ts = []
    for _ in range(n + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[len(ts[1:]) // 2]


for (nb, M, N, K) in [(1, 8192, 8192, 4096), (36, 12544, 512, 512), (36, 12544, 512, 256), (36, 50176, 256, 256), (36, 4096, 512, 512),
                      (36, 200704, 128, 128), (36, 802816, 64, 64)]:
    if nb * M * (K + N) * 4 > 60e9:
        continue
    A = torch.randn(nb, M, K, device=dev)
    B = torch.randn(nb, N, K, device=dev)
    C = torch.empty(nb, M, N, device=dev)
    t = timeit(lambda: lib('gemm_f32_batched_nt', A, K, M * K, B, K, N * K, C, N, M * N, nb, M, N, K, None, 0))
    print(f'batch {nb:2d} M={M:6d} N={N:5d} K={K:5d}: {t:7.3f} ms  {2.0 * nb * M * N * K / t / 1e9:6.1f} TF  '
          f'{4.0 * nb * (M * K + N * K + M * N) / t / 1e9:5.2f} TB/s', flush=True)
    del A, B, C
