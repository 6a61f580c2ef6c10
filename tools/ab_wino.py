#!/usr/bin/env python3
"""A/B the Winograd conv kernel across several builds of the library IN ONE PROCESS, interleaved rounds.
    python tools/ab_wino.py libA.so libB.so ... [--batch 256] [--rounds 5] [--only 28,56]"""
import argparse
import ctypes
import os

import torch

LAYERS = [(224, 64, 64, 1), (112, 64, 128, 0), (112, 128, 128, 1), (56, 128, 256, 0), (56, 256, 256, 0), (28, 256, 512, 0),
          (28, 512, 512, 0), (14, 512, 512, 0)]
COUNT = {(224, 64, 64, 1): 1, (112, 64, 128, 0): 1, (112, 128, 128, 1): 1, (56, 128, 256, 0): 1, (56, 256, 256, 0): 3,
         (28, 256, 512, 0): 1, (28, 512, 512, 0): 3, (14, 512, 512, 0): 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--f43', action='store_true', help='time the F(4x4,3x3) entry points (lctvqa_conv3x3_wino4_*)')
    args = ap.parse_args()
    only = [int(v) for v in args.only.split(',')] if args.only else None
    dev = torch.device('cuda')
    dlls = [ctypes.CDLL(os.path.abspath(p)) for p in args.libs]
    sfx = 'wino4' if args.f43 else 'wino'
    for d in dlls:
        getattr(d, f'lctvqa_conv3x3_{sfx}_packed_elems').restype = ctypes.c_long
    B = args.batch
    P = ctypes.c_void_p
    st = lambda: P(torch.cuda.current_stream().cuda_stream)
    res = {}
    for (H, Cin, Cout, pool) in LAYERS:
        if only and H not in only:
            continue
        if not getattr(dlls[0], f'lctvqa_conv3x3_{sfx}_supported')(B, H, H, Cin, Cout, pool):
            continue
        xin = torch.randn(B, H, H, Cin, device=dev)
        wt = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
        bs = torch.zeros(Cout, device=dev)
        up = torch.empty(getattr(dlls[0], f'lctvqa_conv3x3_{sfx}_packed_elems')(Cout, Cin), device=dev)
        Ho = H // 2 if pool else H
        out = torch.empty(B, Ho, Ho, Cout, device=dev)
        fl = 2.0 * B * H * H * Cout * Cin * 9
        times = [[] for _ in dlls]
        for rnd in range(args.rounds + 1):
            for i, d in enumerate(dlls):
                assert getattr(d, f'lctvqa_conv3x3_{sfx}_pack_weights')(P(wt.data_ptr()), P(up.data_ptr()), Cout, Cin, 0, st()) == 0
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    rc = getattr(d, f'lctvqa_conv3x3_{sfx}_nhwc')(P(xin.data_ptr()), P(up.data_ptr()), P(bs.data_ptr()), P(out.data_ptr()),
                                                    B, H, H, Cin, Cout, 1, pool, st())
                    assert rc == 0
                e1.record(); torch.cuda.synchronize()
                if rnd > 0:
                    times[i].append(e0.elapsed_time(e1) / 3)
        line = f'wino {H:3d} {Cin:3d}->{Cout:3d}{"+p" if pool else "  "}:'
        for i, t in enumerate(times):
            t.sort()
            med = t[len(t) // 2]
            res.setdefault(i, 0.0)
            res[i] += med * COUNT[(H, Cin, Cout, pool)]
            line += f'  [{i}] {med:7.3f} ms {fl / med / 1e9:6.1f} aTF'
        print(line, flush=True)
        del xin, out
    print('weighted total (approx. per step):', {os.path.basename(args.libs[i]): round(v, 3) for i, v in res.items()})


if __name__ == '__main__':
    main()
