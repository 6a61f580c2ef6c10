#!/usr/bin/env python3
"""fp32 GEMM: the LDS-DMA kernel against the general (VGPR-staged) kernel on the shapes of the path, interleaved in one process.
    python tools/bench_gemm.py [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd import ops  # noqa: E402
from lct_vqa_amd._lib import lib  # noqa: E402

# (name, M, N, K, layout) layout: nt = Linear forward, nn = dX, tn = dW
SHAPES = [
    ('vgg fc6 fwd', 256, 4096, 25088, 'nt'), ('vgg fc7 fwd', 256, 4096, 4096, 'nt'), ('img fc fwd', 256, 512, 4096, 'nt'),
    ('fc2 fwd', 256, 3129, 3129, 'nt'), ('token head fwd', 3584, 17858, 512, 'nt'), ('token head dX', 3584, 512, 17858, 'nn'),
    ('token head dW', 17858, 512, 3584, 'tn'),
    ('res 56x56 64->256', 802816, 256, 64, 'nt'), ('res 56x56 256->64', 802816, 64, 256, 'nt'),
    ('res 28x28 128->512', 200704, 512, 128, 'nt'), ('res 28x28 512->128', 200704, 128, 512, 'nt'),
    ('res 14x14 256->1024', 50176, 1024, 256, 'nt'), ('res 14x14 1024->256', 50176, 256, 1024, 'nt'),
    ('res 7x7 512->2048', 12544, 2048, 512, 'nt'), ('res 7x7 2048->512', 12544, 512, 2048, 'nt'),
    ('res 14x14 dX 1024->256', 50176, 256, 1024, 'nn'), ('res 14x14 dW 256x1024', 1024, 256, 50176, 'tn'),
    ('wino plane 28x28 512', 12544, 512, 512, 'nt'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda')
    force = lib.raw('gemm_force_general')
    for name, M, N, K, lay in SHAPES:
        if lay == 'nt':
            A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            call = lambda: ops.gemm(A, K, 1, B, K, 1, M, N, K, out=C)
        elif lay == 'nn':                                          # B(n,k) = W[k][n]
            A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
            call = lambda: ops.gemm(A, K, 1, B, 1, N, M, N, K, out=C)
        else:                                                      # A(m,k) = dY[k][m], B(n,k) = X[k][n]
            A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            call = lambda: ops.gemm(A, 1, M, B, 1, N, M, N, K, out=C)
        C = torch.empty(M, N, device=dev)
        times = {0: [], 1: []}
        for rnd in range(args.rounds + 1):
            for f in (0, 1):
                force(f)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    call()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[f].append(e0.elapsed_time(e1) / 3)
        force(0)
        fl = 2.0 * M * N * K
        by = 4.0 * (M * K + N * K + M * N)
        t0, t1 = sorted(times[0])[len(times[0]) // 2], sorted(times[1])[len(times[1]) // 2]
        print(f'{name:26s} {lay} M={M:7d} N={N:6d} K={K:6d}: dma {t0:7.3f} ms {fl / t0 / 1e9:6.1f} TF {by / t0 / 1e9:6.2f} TB/s | '
              f'general {t1:7.3f} ms {fl / t1 / 1e9:6.1f} TF', flush=True)
        del A, B, C


if __name__ == '__main__':
    main()
