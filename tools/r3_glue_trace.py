#!/usr/bin/env python3
"""Which Python-level ops launch the ATen add / copy / fill kernels of one darts_vqa weight step (torch profiler, grouped by the
operator stack): finds the glue that is left around the fused edges."""
import collections
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

import bench


def main():
    sys.argv = ['bench.py', '--workload', 'darts_vqa', '--batch', '16', '--steps', '1', '--warmup', '1', '--no-cpu-baseline']
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
        bench.main()
    ev = prof.key_averages(group_by_input_shape=True)
    rows = []
    for e in ev:
        if e.key in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::cat', 'aten::clone', 'aten::stack', 'aten::zeros',
                     'aten::mul', 'aten::sum', 'aten::select', 'aten::contiguous', 'aten::_to_copy', 'aten::empty_like'):
            rows.append((e.count, e.key, str(e.input_shapes)[:110], round(e.device_time_total / 1e3, 3)))
    rows.sort(reverse=True)
    for r in rows[:60]:
        print(r)


if __name__ == '__main__':
    main()
