"""A/B: the fused F(4x4,3x3) kernel (conv3x3_wino4_nhwc) against the split path (input transform + 36 batched
fp32 GEMMs + output transform, conv3x3_wino4_split_nhwc) on VGG19's layers at B=256: parity of the two, then time alone on the chip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
B = int(os.environ.get('B', '256'))
layers = [(28, 256, 512, 0), (28, 512, 512, 0), (28, 512, 512, 1), (14, 512, 512, 0), (56, 256, 256, 0), (56, 128, 256, 0)]
def timeit(f, n=5):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for H, Cin, Cout, pool in layers:
    g = torch.Generator().manual_seed(H + Cin)
    x = torch.relu(torch.randn(B, H, H, Cin, generator=g)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    Ho = H // 2 if pool else H
    y0 = torch.empty(B, Ho, Ho, Cout, device=DEV)
    y1 = torch.empty_like(y0)
    up = torch.empty(Cout * Cin * 36, device=DEV)
    lib('conv3x3_wino4_pack_weights', w, up, Cout, Cin, 0)
    U = torch.empty(36 * Cout * Cin, device=DEV)
    lib('conv3x3_wino4_split_pack_weights', w, U, Cout, Cin)
    n = lib.query('conv3x3_wino4_split_workspace_elems', B, H, H, Cin, Cout)
    ws = torch.empty(n, device=DEV)
    fused = lambda: lib('conv3x3_wino4_nhwc', x, up, b, y0, B, H, H, Cin, Cout, 1, pool)
    split = lambda: lib('conv3x3_wino4_split_nhwc', x, U, b, y1, ws, n, B, H, H, Cin, Cout, 1, pool)
    fused(); split(); torch.cuda.synchronize()
    err = ((y0 - y1).abs().max() / y0.abs().max()).item()
    tf, ts = timeit(fused), timeit(split)
    fl = 2.0 * B * H * H * Cout * Cin * 9
    print(f'{H:3d}x{H:<3d} {Cin:4d}->{Cout:<4d} pool {pool}: fused {tf:6.3f} ms ({fl / tf / 1e9:6.1f} TFLOP/s direct-equivalent)   split {ts:6.3f} ms   max rel diff {err:.1e}   workspace {n * 4 / 1e9:.2f} GB')
