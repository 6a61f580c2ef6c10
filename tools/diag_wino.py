#!/usr/bin/env python3
"""Diagnostic: per-workgroup s_memtime stamps of the v2 Winograd kernel (library built with -DWINO_DIAG -DWINO_V3=0).
    python tools/diag_wino.py lib.so H Cin Cout"""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
H, Cin, Cout = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = 256
lib.lctvqa_conv3x3_wino_packed_elems.restype = ctypes.c_long
P = ctypes.c_void_p
dev = torch.device('cuda')
st = lambda: P(torch.cuda.current_stream().cuda_stream)
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
up = torch.empty(lib.lctvqa_conv3x3_wino_packed_elems(Cout, Cin), device=dev)
y = torch.zeros(B, H, H, Cout, device=dev)
assert lib.lctvqa_conv3x3_wino_pack_weights(P(w.data_ptr()), P(up.data_ptr()), Cout, Cin, 0, st()) == 0
for it in range(int(os.environ.get("DIAG_ITERS", "300"))):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.lctvqa_conv3x3_wino_nhwc(P(x.data_ptr()), P(up.data_ptr()), None, P(y.data_ptr()), B, H, H, Cin, Cout, 1, 0, st()) == 0
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
nblk = int(sys.argv[5]) if len(sys.argv) > 5 else (B * H // 8) * (H // 16) * (Cout // 128)
d = y.reshape(-1)[:nblk * 4].reshape(nblk, 4).cpu()
pro, loop, epi = d[:, 0], d[:, 1], d[:, 2]
NCH = Cin // 16
print(f'{ms:.3f} ms; blocks {nblk}; s_memtime ticks per block: prologue {pro.median():.0f}  loop {loop.median():.0f} '
      f'({loop.median() / (NCH * 128):.1f} per MFMA)  epilogue {epi.median():.0f}')
tot = (pro + loop + epi)
rt = d[:, 3]
print(f'  in-kernel clock = {(tot / rt).median() * 0.1:.3f} GHz (s_memtime / s_memrealtime x 100 MHz)')
print('  loop p10/p50/p90:', [float(loop.kthvalue(max(1, int(nblk * q)))[0]) for q in (0.1, 0.5, 0.9)])
print(f'  total per block {tot.median():.0f}; x blocks / 256 CUs = {tot.sum() / 256:.0f} ticks = {tot.sum() / 256 / ms * 1e-6:.3f} GHz-equivalent if tick = shader clock')
