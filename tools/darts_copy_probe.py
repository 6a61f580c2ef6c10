#!/usr/bin/env python3
"""Where do the torch copy / add / fill kernels of a PC-DARTS encoder forward+backward come from?  (torch profiler, grouped by op
and by the python line that issued it)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd.basic_vqa import config  # noqa: E402

config.DEVICE = torch.device('cuda')
from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network  # noqa: E402

torch.manual_seed(0)
net = Network(16, 512, 4).to('cuda').train()
x = torch.randn(16, 3, 224, 224, device='cuda')
params = list(net.parameters()) + net.arch_parameters()
for _ in range(2):
    torch.autograd.grad(net(x).sum(), params)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    torch.autograd.grad(net(x).sum(), params)
torch.cuda.synchronize()
by_op = collections.Counter()
by_site = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::contiguous',
                  'aten::cat', 'aten::empty_strided', 'aten::sum', 'aten::mul'):
        by_op[e.name] += 1
        st = [s for s in (e.stack or []) if 'lct_vqa_amd' in s or 'autograd' in s]
        by_site[(e.name, st[0] if st else '(no python frame: autograd engine)', str(e.input_shapes)[:60])] += 1
print(by_op.most_common())
for (k, n) in by_site.most_common(40):
    print(n, k)
