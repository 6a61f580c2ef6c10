#!/usr/bin/env python3
"""3x3 weight gradient: Winograd-domain F(3x3,4x4) against the direct MFMA kernel, interleaved in one process (B = 256)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd import ops  # noqa: E402
from lct_vqa_amd._lib import lib  # noqa: E402

dev = torch.device('cuda')
B = 256
for (H, Cin, Cout, count) in [(112, 128, 128, 1), (56, 128, 256, 1), (56, 256, 256, 3), (28, 256, 512, 1), (28, 512, 512, 3),
                              (14, 512, 512, 4), (14, 256, 256, 36), (28, 128, 128, 8)]:
    if not lib.query('conv3x3_wino4_wgrad_supported', B, H, H, Cin, Cout):
        print(H, Cin, Cout, 'not supported')
        continue
    x = torch.randn(B, H, H, Cin, device=dev)
    dz = torch.randn(B, H, H, Cout, device=dev)
    times = {'winograd': [], 'direct': []}
    for rnd in range(5):
        for algo in times:
            ops.set_wgrad_algo(algo)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                ops.conv3x3_weight_grad(x, dz, B, H, H, Cin, Cout)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[algo].append(e0.elapsed_time(e1) / 2)
    ops.set_wgrad_algo('winograd')
    fl = 2.0 * B * H * H * Cout * Cin * 9
    tw, td = sorted(times['winograd'])[2], sorted(times['direct'])[2]
    print(f'{H:3d}x{H:<3d} {Cin:3d}->{Cout:3d} (x{count:2d}): winograd {tw:7.3f} ms ({fl / tw / 1e9:6.1f} aTF)  direct {td:7.3f} ms ({fl / td / 1e9:6.1f} TF)',
          flush=True)
    del x, dz
