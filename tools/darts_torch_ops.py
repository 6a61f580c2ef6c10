#!/usr/bin/env python3
"""Which torch-side ops (not this library's kernels) run in one forward+backward of the PC-DARTS encoder: torch.profiler
table grouped by op name, sorted by count.  Diagnostic for the launch-count budget."""
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd.basic_vqa import config  # noqa: E402

config.DEVICE = torch.device('cuda')
from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network  # noqa: E402

torch.manual_seed(0)
net = Network(16, 512, 4).to('cuda').train()
x = torch.randn(4, 3, 64, 64, device='cuda')
params = list(net.parameters()) + net.arch_parameters()
for _ in range(2):
    torch.autograd.grad(net(x).sum(), params)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    torch.autograd.grad(net(x).sum(), params)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 or 'copy' in e.key.lower()]
for k, n, t in sorted(rows, key=lambda r: -r[1])[:40]:
    print(f'{n:6d} {t / 1e3:9.3f} ms  {k[:110]}')
