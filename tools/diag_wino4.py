#!/usr/bin/env python3
"""Per-wave s_memtime stamps of the F(4x4,3x3) kernel (library built with -DW4_DIAG): ticks parked at the end-of-phase
wait + barrier, K-loop ticks, epilogue ticks.   python tools/diag_wino4.py lib.so H Cin Cout"""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
H, Cin, Cout = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = 256
lib.lctvqa_conv3x3_wino4_packed_elems.restype = ctypes.c_long
P = ctypes.c_void_p
dev = torch.device('cuda')
st = lambda: P(torch.cuda.current_stream().cuda_stream)
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
up = torch.empty(lib.lctvqa_conv3x3_wino4_packed_elems(Cout, Cin), device=dev)
y = torch.zeros(B, H, H, Cout, device=dev)
assert lib.lctvqa_conv3x3_wino4_pack_weights(P(w.data_ptr()), P(up.data_ptr()), Cout, Cin, 0, st()) == 0
for it in range(200):
    assert lib.lctvqa_conv3x3_wino4_nhwc(P(x.data_ptr()), P(up.data_ptr()), None, P(y.data_ptr()), B, H, H, Cin, Cout, 1, 0, st()) == 0
torch.cuda.synchronize()
nblk = int(sys.argv[5])
d = y.reshape(-1)[:nblk * 32].reshape(nblk, 8, 4).cpu()
d = d[d[:, 0, 1] > 0]                                            # workgroups that ran (the XCD map leaves some ids idle)
NCH = Cin // 8
print(f'blocks {len(d)}; per wave (median over blocks): wait+barrier ticks / loop ticks / epilogue ticks; loop per MFMA = loop/(NCH*36)')
for wv in range(8):
    wt, lp, ep = d[:, wv, 0].median().item(), d[:, wv, 1].median().item(), d[:, wv, 2].median().item()
    print(f'  wave {wv}: {wt:9.0f} ({100 * wt / lp:4.1f} % of loop)  {lp:9.0f} ({lp / (NCH * 36):5.1f} per MFMA)  {ep:8.0f}')
