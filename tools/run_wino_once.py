#!/usr/bin/env python3
"""Run one Winograd conv layer a few times (profiling target).  python tools/run_wino_once.py {wino|wino4} H Cin Cout [iters]"""
import sys
import torch
sys.path.insert(0, '.')
from lct_vqa_amd._lib import lib
v, H, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
B = 256
dev = torch.device('cuda')
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
b = torch.zeros(Cout, device=dev)
up = torch.empty(lib.query(f'conv3x3_{v}_packed_elems', Cout, Cin), device=dev)
lib(f'conv3x3_{v}_pack_weights', w, up, Cout, Cin, 0)
y = torch.empty(B, H, H, Cout, device=dev)
for _ in range(it):
    lib(f'conv3x3_{v}_nhwc', x, up, b, y, B, H, H, Cin, Cout, 1, 0)
torch.cuda.synchronize()
