#!/usr/bin/env python3
"""Are two builds of the fused-edge FORWARD bit-identical?  (the matrix-core forward kernels against the first form, -DEDGE_FWD_V1)
    python tools/ab_edge_exact.py libA.so libB.so"""
import ctypes
import os
import sys

import torch

SHAPES = [(32, 1, 2, 112, 112), (32, 2, 2, 224, 224), (64, 1, 2, 56, 56), (64, 2, 2, 112, 112), (64, 1, 1, 18, 50), (32, 2, 1, 20, 68), (64, 2, 3, 36, 44),
          (32, 1, 3, 9, 33)]


def main():
    dev = torch.device('cuda')
    dlls = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
    P_ = ctypes.c_void_p
    st = lambda: P_(torch.cuda.current_stream().cuda_stream)
    worst = 0
    for (C, s, B, H, W) in SHAPES:
        torch.manual_seed(C + s + H)
        Cq, Ho, Wo = C // 4, H // s, W // s
        Pn = B * Ho * Wo
        x = torch.randn(B, H, W, C, device=dev)
        ws = [torch.randn(Cq, 1, k, k, device=dev) * 0.2 for k in (3, 5, 3, 5, 3, 5)] + [torch.randn(Cq, Cq, device=dev) * 0.3 for _ in range(6)]
        if s == 2:
            ws += [torch.randn(Cq // 2, Cq, device=dev) * 0.3 for _ in range(2)]
        wptr = (ctypes.c_void_p * 14)(*([w.data_ptr() for w in ws] + [None] * (14 - len(ws))))
        bn = (ctypes.c_void_p * 27)(*([None] * 27))
        wsel = torch.softmax(torch.randn(7, device=dev), 0)
        beta = torch.ones(1, device=dev)
        outs = []
        for d in dlls:
            grid = d.lctvqa_mixed_edge_grid(B, H, W, C, s)
            pre = torch.full((9, Pn, Cq), float('nan'), device=dev)
            stats = torch.full((2, 9, Cq), float('nan'), device=dev)
            part = torch.empty(grid * 7 * Cq * 2 + 51 * Cq, dtype=torch.float64, device=dev)
            out = torch.empty(B, Ho, Wo, C, device=dev)
            rc = d.lctvqa_mixed_edge_fwd(P_(x.data_ptr()), B, H, W, C, s, wptr, bn, P_(pre.data_ptr()), P_(stats[0].data_ptr()), P_(stats[1].data_ptr()),
                                         P_(part.data_ptr()), P_(wsel.data_ptr()), P_(beta.data_ptr()), P_(out.data_ptr()), ctypes.c_long(C), 0, 1,
                                         ctypes.c_float(0.1), ctypes.c_float(1e-5), None, st())
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs.append((pre, stats, out))
        (p0, s0, o0), (p1, s1, o1) = outs
        live = [0, 1, 3, 4, 5, 6, 7, 8] + ([2] if s == 2 else [])
        line = f'C={C} s={s} B={B} {H}x{W}:'
        for k in live:
            same = torch.equal(p0[k], p1[k])
            d = float((p0[k] - p1[k]).abs().max())
            line += f' pre{k}={"=" if same else f"{d:.2e}"}'
            worst += 0 if same else 1
        for nm, a, b in (('mean', s0[0], s1[0]), ('invstd', s0[1], s1[1]), ('out', o0, o1)):
            same = torch.equal(a[~torch.isnan(a)], b[~torch.isnan(b)])
            line += f' {nm}={"=" if same else f"{float((a - b).abs().nan_to_num().max()):.2e}"}'
            worst += 0 if same else 1
        print(line, flush=True)
    print('tensors that differ:', worst)
    sys.exit(1 if worst else 0)


if __name__ == '__main__':
    main()
