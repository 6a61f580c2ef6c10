#!/usr/bin/env python3
"""Time the fused MixedOp edge forward / backward C calls across several builds of the library in one process (ablation builds of
csrc/mixed_edge.hip: tools/build_one_variant.sh mixed_edge out.so -DEDGE_ABL_...).  Results of ablation builds are wrong by design;
only the timings are read.
    python tools/ab_edge.py libA.so libB.so ... [--batch 64]"""
import argparse
import ctypes
import os

import torch

SHAPES = [(16, 1, 224), (32, 2, 224), (32, 1, 112), (64, 2, 112), (64, 1, 56)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=4)
    args = ap.parse_args()
    dev = torch.device('cuda')
    dlls = [ctypes.CDLL(os.path.abspath(p)) for p in args.libs]
    for d in dlls:
        d.lctvqa_mixed_edge_bwd_scratch_floats.restype = ctypes.c_long
    P_ = ctypes.c_void_p
    B = args.batch
    st = lambda: P_(torch.cuda.current_stream().cuda_stream)
    tot = [[0.0, 0.0] for _ in dlls]
    for (C, s, H) in SHAPES:
        Cq, Ho = C // 4, H // s
        Pn = B * Ho * Ho
        x = torch.randn(B, H, H, C, device=dev)
        ws = [torch.randn(Cq, 1, k, k, device=dev) * 0.2 for k in (3, 5, 3, 5, 3, 5)] + [torch.randn(Cq, Cq, device=dev) * 0.3 for _ in range(6)]
        if s == 2:
            ws += [torch.randn(Cq // 2, Cq, device=dev) * 0.3 for _ in range(2)]
        wptr = (ctypes.c_void_p * 14)(*([w.data_ptr() for w in ws] + [None] * (14 - len(ws))))
        bn = (ctypes.c_void_p * 27)(*([None] * 27))
        pre = torch.empty(9, Pn, Cq, device=dev)
        stats = torch.empty(2, 9, Cq, device=dev)
        grid = max(d.lctvqa_mixed_edge_grid(B, H, H, C, s) for d in dlls)
        part = torch.empty(grid * 7 * Cq * 2 + 51 * Cq, dtype=torch.float64, device=dev)
        wsel = torch.softmax(torch.randn(7, device=dev), 0)
        beta = torch.ones(1, device=dev)
        out = torch.empty(B, Ho, Ho, C, device=dev)
        dout = torch.randn(B, Ho, Ho, C, device=dev)
        dx = torch.empty_like(x)
        dw, dbeta = torch.empty(7, device=dev), torch.empty(1, device=dev)
        wgrad = torch.empty(max(d.lctvqa_mixed_edge_wgrad_elems(C) for d in dlls), device=dev)
        scratch = torch.empty(max(d.lctvqa_mixed_edge_bwd_scratch_floats(B, H, H, C, s) for d in dlls), device=dev)   # builds differ in their scratch layout
        times = [[[], []] for _ in dlls]
        for rnd in range(args.rounds + 1):
            for i, d in enumerate(dlls):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                for _ in range(3):
                    rc = d.lctvqa_mixed_edge_fwd(P_(x.data_ptr()), B, H, H, C, s, wptr, bn, P_(pre.data_ptr()), P_(stats[0].data_ptr()), P_(stats[1].data_ptr()),
                                                 P_(part.data_ptr()), P_(wsel.data_ptr()), P_(beta.data_ptr()), P_(out.data_ptr()), ctypes.c_long(C), 0, 1,
                                                 ctypes.c_float(0.1), ctypes.c_float(1e-5), None, st())
                    assert rc == 0
                ev[1].record()
                for _ in range(3):
                    rc = d.lctvqa_mixed_edge_bwd(P_(dout.data_ptr()), ctypes.c_long(C), P_(x.data_ptr()), B, H, H, C, s, wptr, P_(pre.data_ptr()),
                                                 P_(stats[0].data_ptr()), P_(stats[1].data_ptr()), P_(wsel.data_ptr()), P_(beta.data_ptr()), P_(dx.data_ptr()),
                                                 P_(dw.data_ptr()), P_(dbeta.data_ptr()), P_(wgrad.data_ptr()), P_(scratch.data_ptr()), 1, 0, None, 1, st())
                    assert rc == 0
                ev[2].record()
                torch.cuda.synchronize()
                if rnd:
                    times[i][0].append(ev[0].elapsed_time(ev[1]) / 3)
                    times[i][1].append(ev[1].elapsed_time(ev[2]) / 3)
        T = 4.0 * Pn * Cq
        line = f'C={C:2d} s={s} {H:3d}^2 B={B}:'
        for i in range(len(dlls)):
            f, b = (sorted(t)[len(t) // 2] for t in times[i])
            tot[i][0] += f; tot[i][1] += b
            line += f'  [{i}] fwd {f * 1e3:7.1f} us ({29 * T / f / 1e9:5.2f} TB/s alg) bwd {b * 1e3:7.1f} us ({60 * T / b / 1e9:5.2f} TB/s alg)'
        print(line, flush=True)
    print('sum over the five edge shapes (ms):', {os.path.basename(args.libs[i]): [round(v, 3) for v in tot[i]] for i in range(len(dlls))})


if __name__ == '__main__':
    main()
