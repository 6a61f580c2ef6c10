#!/usr/bin/env python3
"""Time the LSTM recurrence alone (B=256, H=512, T=14: the question encoder of BASELINE configs[1]):
the packed whole-sequence entry points against the per-step kernels they replace.
    python tools/bench_lstm.py [--B 256] [--H 512] [--T 14]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lct_vqa_amd._lib import lib  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=256)
    ap.add_argument('--H', type=int, default=512)
    ap.add_argument('--T', type=int, default=14)
    a = ap.parse_args()
    B, H, T = a.B, a.H, a.T
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    xp, w_hh, b_hh = r(T, B, 4 * H), r(4 * H, H) * H ** -0.5, r(4 * H) * 0.1
    h_all, c_all = torch.zeros(T + 1, B, H, device=dev), torch.zeros(T + 1, B, H, device=dev)
    gates = torch.empty(T, B, 4 * H, device=dev)
    wpack = torch.empty(lib.query('lstm_seq_scratch_elems', H), device=dev)
    dgates, dout = r(T, B, 4 * H), r(T, B, H)
    dc_buf = torch.zeros(2, B, H, device=dev)

    def fwd_steps():
        for t in range(T):
            lib('lstm_step_fused', xp[t], h_all[t], c_all[t], w_hh, b_hh, gates[t], c_all[t + 1], h_all[t + 1], B, H)

    def fwd_seq():
        lib('lstm_seq_fwd', xp, w_hh, b_hh, h_all, c_all, gates, wpack, T, B, H)

    def bwd_steps():
        for t in range(T - 2, -1, -1):
            lib('lstm_step_bwd_fused', dgates[t + 1], w_hh, dout[t], dc_buf[t & 1], gates[t], c_all[t], c_all[t + 1], dgates[t],
                dc_buf[(t + 1) & 1], B, H)

    def bwd_seq():
        lib('lstm_seq_bwd', dgates, w_hh, dout, gates, c_all, dc_buf, wpack, T, B, H)

    fwd_seq()
    ref_h = h_all.clone()
    fwd_steps()
    print('fwd per-step vs packed max diff', (h_all - ref_h).abs().max().item())
    flop_f = 2.0 * T * B * 4 * H * H
    for name, fn, fl in (('fwd per-step kernels', fwd_steps, flop_f), ('fwd packed sequence ', fwd_seq, flop_f),
                         ('bwd per-step kernels', bwd_steps, flop_f * (T - 1) / T), ('bwd packed sequence ', bwd_seq, flop_f * (T - 1) / T)):
        us = timed(fn)
        print(f'{name}: {us:8.1f} us per sequence = {us / T:6.1f} us per step, {fl / us * 1e-6:6.1f} TFLOP/s')


if __name__ == '__main__':
    main()
