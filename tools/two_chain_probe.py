#!/usr/bin/env python3
"""Does running the frozen fp32 trunk as two half-batch chains on two streams fill the grid tails of its layers?
    python tools/two_chain_probe.py [--batch 256] [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd import ops  # noqa: E402

CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--chains', type=int, default=2)
    args = ap.parse_args()
    dev = torch.device('cuda')
    torch.manual_seed(0)
    convs, cin = [], 3
    for c in CFG:
        if c == 'M':
            convs.append('M')
        else:
            convs.append((torch.randn(c, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5, torch.zeros(c, device=dev)))
            cin = c
    B = args.batch
    img = torch.randn(B, 3, 224, 224, device=dev)
    streams = [torch.cuda.Stream() for _ in range(args.chains)]
    parts = img.chunk(args.chains)

    def single():
        return ops.vgg_features_forward(img, convs, cache=True)[0]

    def chained():
        cur = torch.cuda.current_stream()
        outs = []
        for s, p in zip(streams, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(ops.vgg_features_forward(p, convs, cache=True)[0])
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)

    ref = single()
    got = chained()
    torch.cuda.synchronize()
    print('max |single - chained| =', (ref - got).abs().max().item(), 'of', ref.abs().max().item())
    for name, fn in (('single', single), ('chained', chained), ('single', single), ('chained', chained)):
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f'{name:8s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}', flush=True)


if __name__ == '__main__':
    main()
