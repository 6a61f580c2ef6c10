"""Time the fp32 NT GEMM kernel on exactly n x 256 workgroups (256 x 128 tiles, K = 512): how much do co-resident workgroups overlap?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
K, N = 512, 128
def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for wgs in (64, 128, 256, 512, 768, 1024, 2048, 4096):
    T = 256 * wgs
    V = torch.randn(T, K, device=DEV); U = torch.randn(N, K, device=DEV); M = torch.empty(T, N, device=DEV)
    t = timeit(lambda: lib('gemm_f32_batched', V, K, 1, 0, U, K, 1, 0, M, N, 0, 1, T, N, K, None, 0))
    print(f'{wgs:5d} workgroups: {t * 1e3:8.1f} us   {2.0 * T * N * K / t / 1e9:7.1f} TFLOP/s   per-WG MFMA time at 2.4 GHz: {32 * 4096 / 2.4e3:.1f} us', flush=True)
