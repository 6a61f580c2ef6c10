"""Per-shape timing of conv_gemm_bf16 (csrc/train_bf16.hip) on the 1x1 products of ResNet-152 at B=256: forward (+ statistics),
data gradient with ReLU mask, data gradient with residual addend.  LCTVQA_CG_ROWS=128|256 forces the tile height."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
B = int(os.environ.get('B', '256'))
BF = torch.bfloat16
shapes = [(56, 256, 64), (56, 64, 256), (28, 512, 128), (28, 128, 512), (14, 1024, 256), (14, 256, 1024), (7, 2048, 512), (7, 512, 2048)]
tot = 0.0
for H, K, N in shapes:
    P = B * H * H
    x = torch.randn(P, K, device=DEV).to(BF)
    w = torch.randn(N, K, device=DEV).to(BF)
    y = torch.empty(P, N, dtype=BF, device=DEV)
    aux = torch.randn(P, N, device=DEV).to(BF)
    slabs = lib.query('conv_gemm_bf16_stats_slabs', P, N, K, 1)
    st = torch.empty(slabs * 2 * N, device=DEV)
    for tag, add, msk, stats in (('fwd+stats', None, None, st), ('dgrad+mask', None, aux, None), ('dgrad+addend', aux, None, None)):
        f = lambda: lib('conv_gemm_bf16', x, K, w, K, y, N, P, N, K, 1, 0, 0, add, N, msk, N, stats)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        by = 2.0 * P * (K + N + (N if (add is not None or msk is not None) else 0))
        tot += ms
        print(f'{H:3d}x{H:<3d} {K:5d}->{N:<5d} {tag:13s}: {ms * 1e3:8.1f} us  {by / ms / 1e9:6.2f} TB/s  {2.0 * P * K * N / ms / 1e9:7.1f} TFLOP/s')
print('sum', tot)
