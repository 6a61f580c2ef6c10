"""Time the three kernels of the split F(4x4,3x3) path separately (28x28 512->512, B=256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
B, H, Cin, Cout = 256, int(os.environ.get('H', '28')), 512, 512
T = B * ((H + 3) // 4) ** 2
x = torch.randn(B, H, H, Cin, device=DEV)
V = torch.randn(36, T, Cin, device=DEV)
U = torch.randn(36, Cout, Cin, device=DEV)
M = torch.empty(36, T, Cout, device=DEV)
def timeit(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = lambda: lib('gemm_f32_batched', V, Cin, 1, T * Cin, U, Cin, 1, Cout * Cin, M, Cout, T * Cout, 36, T, Cout, Cin, None, 0)
t = timeit(g)
print(f'batched GEMM 36 x [{T} x {Cin}] x [{Cin} x {Cout}]: {t:.3f} ms = {36 * 2.0 * T * Cin * Cout / t / 1e9:.1f} TFLOP/s ({36 * 4.0 * T * (Cin + Cout) / t / 1e6:.0f} GB/s of operands)')
want = V[5].double() @ U[5].double().T
print('check batch 5 rel err', float((M[5].double() - want).norm() / want.norm()))
