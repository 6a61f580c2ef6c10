#!/usr/bin/env python3
"""Time the 1x1-conv weight gradient for small channel counts at the PC-DARTS shapes (B=16, 224x224 input) and check it
against torch on the same data.
    python tools/bench_pw_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd._lib import lib  # noqa: E402

#         B   H   Cout Cin
SHAPES = [(16, 224, 4, 4), (16, 112, 8, 8), (16, 56, 16, 16), (16, 112, 4, 4), (16, 56, 8, 8), (16, 224, 16, 48), (16, 112, 32, 64),
          (16, 224, 8, 4), (16, 56, 4, 16), (3, 7, 16, 16)]


def main():
    dev = torch.device('cuda')
    for (B, H, Cout, Cin) in SHAPES:
        P = B * H * H
        x = torch.randn(P, Cin, device=dev)
        dy = torch.randn(P, Cout, device=dev)
        dw = torch.empty(Cout, Cin, device=dev)
        if not lib.query('pwconv_wgrad_small_supported', P, Cout, Cin):
            print(f'P={P} {Cin}->{Cout}: unsupported')
            continue
        scratch = torch.empty(lib.query('pwconv_wgrad_small_scratch_elems', P, Cout, Cin), device=dev)
        fn = lambda: lib('pwconv_wgrad_small', dy, x, dw, scratch, P, Cout, Cin, 1)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        ref = dy.double().t() @ torch.relu(x).double()
        err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
        mb = (x.numel() + dy.numel()) * 4 / 1e6
        print(f'P={P:7d} {Cin:2d}->{Cout:2d}: {us:6.1f} us  ({mb:5.1f} MB streamed, {mb / us:5.2f} TB/s)  rel err {err:.1e}')


if __name__ == '__main__':
    main()
