#!/usr/bin/env python3
"""A/B the bf16 conv kernel across several builds of the library IN ONE PROCESS, interleaved rounds.
    python tools/ab_conv_bf16.py libA.so libB.so ... [--batch 256] [--rounds 5]"""
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
    args = ap.parse_args()
    dev = torch.device('cuda')
    dlls = [ctypes.CDLL(os.path.abspath(p)) for p in args.libs]
    B = args.batch
    P = ctypes.c_void_p
    st = lambda: P(torch.cuda.current_stream().cuda_stream)
    res, outs = {}, {}
    for (H, Cin, Cout, pool) in LAYERS:
        xin = torch.randn(B, H, H, Cin, device=dev).bfloat16()
        wt = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
        bs = torch.zeros(Cout, device=dev)
        wp = torch.empty(Cout * Cin * 9, device=dev, dtype=torch.bfloat16)
        Ho = H // 2 if pool else H
        fl = 2.0 * B * H * H * Cout * Cin * 9
        times = [[] for _ in dlls]
        ref = None
        for rnd in range(args.rounds + 1):
            for i, d in enumerate(dlls):
                out = torch.empty(B, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
                assert d.lctvqa_conv3x3_bf16_pack_weights(P(wt.data_ptr()), P(wp.data_ptr()), Cout, Cin, st()) == 0
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    rc = d.lctvqa_conv3x3_bf16_nhwc(P(xin.data_ptr()), P(wp.data_ptr()), P(bs.data_ptr()), P(out.data_ptr()),
                                                    B, H, H, Cin, Cout, 1, pool, st())
                    assert rc == 0
                e1.record(); torch.cuda.synchronize()
                if rnd > 0:
                    times[i].append(e0.elapsed_time(e1) / 3)
                if rnd == 0:
                    if ref is None:
                        ref = out.float()
                    else:
                        assert torch.equal(ref, out.float()), f'build {i} differs from build 0 on layer {(H, Cin, Cout, pool)}'
                del out
        line = f'bf16 {H:3d} {Cin:3d}->{Cout:3d}{"+p" if pool else "  "}:'
        for i, t in enumerate(times):
            t.sort()
            med = t[len(t) // 2]
            res.setdefault(i, 0.0)
            res[i] += med * COUNT[(H, Cin, Cout, pool)]
            line += f'  [{i}] {med:7.3f} ms {fl / med / 1e9:6.1f} TF'
        print(line, flush=True)
        del xin
    print('weighted total (approx. per step):', {os.path.basename(args.libs[i]): round(v, 3) for i, v in res.items()})


if __name__ == '__main__':
    main()
