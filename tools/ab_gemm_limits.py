"""Which resource caps the batched fp32 NT GEMM of the split F(4x4,3x3) path: operands from HBM vs. resident, random vs. zero data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
T, Cin, Cout, NB = int(os.environ.get('T', '12544')), 512, 512, 36
def timeit(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, zero, sa, sc in (('random, streamed', False, 1, 1), ('random, A and C shared by the batch', False, 0, 0), ('random, C shared', False, 1, 0),
                           ('random, A shared', False, 0, 1), ('zeros, streamed', True, 1, 1), ('zeros, shared', True, 0, 0)):
    V = torch.zeros(NB, T, Cin, device=DEV) if zero else torch.randn(NB, T, Cin, device=DEV)
    U = torch.zeros(NB, Cout, Cin, device=DEV) if zero else torch.randn(NB, Cout, Cin, device=DEV)
    M = torch.empty(NB, T, Cout, device=DEV)
    t = timeit(lambda: lib('gemm_f32_batched', V, Cin, 1, sa * T * Cin, U, Cin, 1, Cout * Cin, M, Cout, sc * T * Cout, NB, T, Cout, Cin, None, 0))
    print(f'{name:40s} {t:.3f} ms = {NB * 2.0 * T * Cin * Cout / t / 1e9:.1f} TFLOP/s', flush=True)
    del V, U, M
