"""Per-shape timing of the bf16 weight-gradient product (csrc/train_bf16.hip) on the ResNet-152 shapes at B=256.
LCTVQA_WGRAD_TWO_STAGE=1 selects the two-stage form.  Prints ms, TB/s of unique operand bytes and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
B = int(os.environ.get('B', '256'))
shapes = [  # (H, Cout, Cin, taps)
    (56, 64, 256, 1), (56, 256, 64, 1), (56, 64, 64, 9), (28, 128, 512, 1), (28, 512, 128, 1), (28, 128, 128, 9),
    (14, 256, 1024, 1), (14, 1024, 256, 1), (14, 256, 256, 9), (7, 512, 2048, 1), (7, 2048, 512, 1), (7, 512, 512, 9)]
for H, Cout, Cin, taps in shapes:
    P = B * H * H
    dz = torch.randn(P, Cout, device=DEV).to(torch.bfloat16)
    x = torch.randn(P, Cin, device=DEV).to(torch.bfloat16)
    need = lib.query('wgrad_bf16_workspace_bytes', P, Cout, Cin, taps, H, H)
    ws = torch.empty(need // 4 + 1, device=DEV)
    dW = torch.empty(Cout * Cin * taps, device=DEV)
    for _ in range(3):
        lib('wgrad_bf16', dz, Cout, x, Cin, P, Cout, Cin, taps, H, H, dW, 0, ws, need)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        lib('wgrad_bf16', dz, Cout, x, Cin, P, Cout, Cin, taps, H, H, dW, 0, ws, need)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'{H:3d}x{H:<3d} {Cin:5d}->{Cout:<5d} taps {taps}: {ms * 1e3:8.1f} us  {2.0 * P * (Cout + Cin) / ms / 1e9:6.2f} TB/s unique  '
          f'{2.0 * P * Cout * Cin * taps / ms / 1e9:7.1f} TFLOP/s  slabs {need / 1e6:6.1f} MB')
