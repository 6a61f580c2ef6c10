#!/usr/bin/env python3
"""Winograd kernel vs torch conv2d on the small shapes of the trainable-trunk tests (fwd and dgrad-style calls)."""
import itertools, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from lct_vqa_amd._lib import lib
dev = torch.device('cuda')
torch.manual_seed(0)
bad = 0
PFX = 'conv3x3_wino4' if (len(sys.argv) > 1 and sys.argv[1] == '4') else 'conv3x3_wino'
TOL = 2e-5 if PFX.endswith('4') else 1e-4
for B, H, (Cin, Cout), relu, pool, use_bias, flip in itertools.product(
        [2, 3], [8, 16, 20, 28, 32, 40, 56], [(64, 64), (64, 128), (128, 64), (128, 128), (256, 128), (512, 512)], [0, 1], [0, 1], [0, 1], [0, 1]):
    W = H
    if not lib.query(PFX + '_supported', B, H, W, Cin, Cout, pool):
        continue
    x = torch.randn(B, H, W, Cin, device=dev)
    # forward-style weight [Cout][Cin][3][3]; flip: w is the FORWARD parameter [Cin][Cout][3][3] of the layer whose dgrad this is
    if flip:
        wf = torch.randn(Cin, Cout, 3, 3, device=dev) * 0.05
        weff = wf.flip(2, 3).transpose(0, 1).contiguous()
    else:
        wf = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        weff = wf
    b = torch.randn(Cout, device=dev) if use_bias else None
    up = torch.empty(lib.query(PFX + '_packed_elems', Cout, Cin), device=dev)
    lib(PFX + '_pack_weights', wf, up, Cout, Cin, flip)
    Ho = H // 2 if pool else H
    y = torch.full((B, Ho, Ho, Cout), float('nan'), device=dev)
    lib(PFX + '_nhwc', x, up, b, y, B, H, W, Cin, Cout, relu, pool)
    ref = F.conv2d(x.permute(0, 3, 1, 2), weff, b, padding=1)
    if relu: ref = torch.relu(ref)
    if pool: ref = F.max_pool2d(ref, 2, 2)
    ref = ref.permute(0, 2, 3, 1)
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    worst = max(globals().get('worst', 0.0), err)
    if not err < TOL:
        bad += 1
        print(f'BAD B={B} H={H} {Cin}->{Cout} relu={relu} pool={pool} bias={use_bias} flip={flip}: {err:.3e}')
print('bad cases:', bad, 'worst', worst)
