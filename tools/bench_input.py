#!/usr/bin/env python3
"""PCIe-inclusive rate of the input pipeline (row f-1): uint8 images from pinned host memory -> HBM -> fp32 NCHW,
B=256 at 224x224, and the fp32-over-PCIe alternative the reference uses.   python tools/bench_input.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd._lib import lib  # noqa: E402
from lct_vqa_amd.data import MEAN, STD  # noqa: E402


def main():
    lib.load()
    B, S = 256, 224
    h8 = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).pin_memory()
    hf = torch.randn(B, 3, S, S).pin_memory()
    out = torch.empty(B, 3, S, S, device='cuda')
    for name, fn in (('uint8 H2D + GPU normalise', lambda: lib('image_u8_nhwc_to_f32_nchw', h8.to('cuda', non_blocking=True), out, B, S, S, *MEAN, *STD)),
                     ('fp32 H2D (reference layout)', lambda: out.copy_(hf, non_blocking=True))):
        for it in range(6):
            if it == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f'{name}: {dt * 1e3:.2f} ms per batch of {B} -> {B / dt:.0f} images/s')


if __name__ == '__main__':
    main()
