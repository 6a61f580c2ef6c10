#!/usr/bin/env python3
"""Time one forward+backward of the PC-DARTS encoder (Network(16,512,4)) and of a darts_vqa train step.
    python tools/bench_darts.py [--batch 16] [--size 224]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--iters', type=int, default=3)
    args = ap.parse_args()
    from lct_vqa_amd.basic_vqa import config
    config.DEVICE = torch.device('cuda')
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
    torch.manual_seed(0)
    net = Network(16, 512, 4).to('cuda').train()
    x = torch.randn(args.batch, 3, args.size, args.size, device='cuda')
    for it in range(args.iters + 1):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = net(x)
        g = torch.autograd.grad(out.sum(), list(net.parameters()) + net.arch_parameters())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    macs = 1408.9e6 * (args.size / 224.0) ** 2
    print(f'DARTS encoder fwd+bwd B={args.batch} {args.size}x{args.size}: {dt * 1e3:.1f} ms  '
          f'({args.batch / dt:.1f} img/s, ~{3 * 2 * macs * args.batch / dt / 1e12:.2f} TFLOP/s, peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB)')


if __name__ == '__main__':
    main()
