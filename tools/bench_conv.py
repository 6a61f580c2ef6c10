#!/usr/bin/env python3
"""Micro-benchmark of the conv3x3 kernels on the VGG19 layer shapes at batch B (random data).
    python tools/bench_conv.py [--batch 256] [--iters 5] [--only 224,14]
Prints ms and TFLOP/s per layer (HIP events on the launch stream) and the total."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd._lib import lib  # noqa: E402

LAYERS = [(224, 64, 64, 1), (112, 64, 128, 0), (112, 128, 128, 1), (56, 128, 256, 0), (56, 256, 256, 0), (56, 256, 256, 0),
          (56, 256, 256, 1), (28, 256, 512, 0), (28, 512, 512, 0), (28, 512, 512, 0), (28, 512, 512, 1),
          (14, 512, 512, 0), (14, 512, 512, 0), (14, 512, 512, 0), (14, 512, 512, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--bwd', action='store_true', help='also time wgrad and dgrad')
    ap.add_argument('--wino', action='store_true', help='also time the Winograd F(2x2,3x3) kernel')
    args = ap.parse_args()
    lib.load()
    dev = torch.device('cuda')
    B = args.batch
    only = [int(v) for v in args.only.split(',')] if args.only else None
    tot_ms = tot_fl = 0.0
    # first layer
    x = torch.randn(B, 3, 224, 224, device=dev)
    w = torch.randn(64, 3, 3, 3, device=dev) * 0.2
    b = torch.zeros(64, device=dev)
    y = torch.empty(B, 224, 224, 64, device=dev)
    for it in range(args.iters + 1):
        if it == 1:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        lib('conv3x3_first_nchw', x, w, b, y, B, 224, 224, 64, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f'first 3->64 @224: {ms:.3f} ms  ({y.numel() * 4 / ms / 1e6:.0f} GB/s written)')
    del x, y
    seen = {}
    wino_ms = {}
    for (H, Cin, Cout, pool) in LAYERS:
        if only and H not in only:
            continue
        key = (H, Cin, Cout, pool)
        if key not in seen:
            xin = torch.randn(B, H, H, Cin, device=dev)
            wt = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
            bs = torch.zeros(Cout, device=dev)
            wp = torch.empty(Cout * Cin * 9, device=dev)
            lib('conv3x3_pack_weights', wt, wp, Cout, Cin)
            Ho = H // 2 if pool else H
            out = torch.empty(B, Ho, Ho, Cout, device=dev)
            for it in range(args.iters + 1):
                if it == 1:
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                lib('conv3x3_nhwc', xin, wt, wp, bs, out, None, B, H, H, Cin, Cout, 1, pool)
            e1.record(); torch.cuda.synchronize()
            seen[key] = e0.elapsed_time(e1) / args.iters
            if args.wino and lib.query('conv3x3_wino_supported', B, H, H, Cin, Cout, pool):
                up = torch.empty(Cout * Cin * 16, device=dev)
                lib('conv3x3_wino_pack_weights', wt, up, Cout, Cin, 0)
                for it in range(args.iters + 1):
                    if it == 1:
                        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                    lib('conv3x3_wino_nhwc', xin, up, bs, out, B, H, H, Cin, Cout, 1, pool)
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / args.iters
                wino_ms[key] = t
                print(f'   wino  {H}x{H} {Cin}->{Cout}: {t:7.3f} ms  {2.0 * B * H * H * Cout * Cin * 9 / t / 1e9:6.1f} algorithmic TFLOP/s '
                      f'({2.0 * B * H * H * Cout * Cin * 4 / t / 1e9:6.1f} executed)')
            if args.bwd:
                dz = torch.randn(B, H, H, Cout, device=dev)
                dw = torch.empty_like(wt)
                ns = lib.query('conv3x3_wgrad_scratch_elems', B, H, H, Cin, Cout)
                scratch = torch.empty(max(ns, 1), device=dev)
                dx = torch.empty(B, H, H, Cin, device=dev)
                wpd = torch.empty(Cout * Cin * 9, device=dev)
                lib('conv3x3_pack_weights_dgrad', wt, wpd, Cout, Cin)
                for nm, fn in (('wgrad', lambda: lib('conv3x3_wgrad_nhwc', xin, dz, dw, scratch, B, H, H, Cin, Cout)),
                               ('dgrad', lambda: lib('conv3x3_nhwc', dz, wt, wpd, None, dx, None, B, H, H, Cout, Cin, 0, 0))):
                    for it in range(args.iters + 1):
                        if it == 1:
                            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                        fn()
                    e1.record(); torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) / args.iters
                    print(f'   {nm} {H}x{H} {Cin}->{Cout}: {t:7.3f} ms  {2.0 * B * H * H * Cout * Cin * 9 / t / 1e9:6.1f} TFLOP/s')
                del dz, dx
            del xin, out
        ms = seen[key]
        fl = 2.0 * B * H * H * Cout * Cin * 9
        tot_ms += ms; tot_fl += fl
        print(f'conv {H:3d}x{H:<3d} {Cin:3d}->{Cout:3d}{"+pool" if pool else "     "}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s')
    if wino_ms:
        tw = sum(wino_ms.get((H, Cin, Cout, pool), seen[(H, Cin, Cout, pool)]) for (H, Cin, Cout, pool) in LAYERS if not only or H in only)
        print(f'TOTAL winograd: {tw:.3f} ms  {tot_fl / tw / 1e9:.1f} algorithmic TFLOP/s')
    print(f'TOTAL mfma conv: {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s  ({tot_fl / tot_ms / 1e9 / 157.3 * 100:.1f}% of 157.3)')


if __name__ == '__main__':
    main()
