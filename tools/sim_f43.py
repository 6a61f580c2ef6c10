#!/usr/bin/env python3
"""Numerical feasibility of Winograd F(4x4,3x3) in fp32 (CPU simulation with explicit fp32 transforms): single-layer
error against an fp64 direct convolution, for the trunk's layer shapes and activation statistics."""
import torch
import torch.nn.functional as F
torch.manual_seed(0)
# Lavin & Gray F(4x4,3x3)
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                   [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)


def wino43(x, w, chunk=16):
    """x [B,C,H,W] (H,W % 4 == 0), w [K,C,3,3]; fp32 throughout; channel accumulation in chunks like the kernel."""
    B, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 6, 4).unfold(3, 6, 4)                       # [B,C,th,tw,6,6]
    V = torch.einsum('ij,bcxyjk,lk->bcxyil', BT, tiles, BT)           # B^T d B
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)                     # G g G^T   [K,C,6,6]
    M = torch.zeros(B, K, V.shape[2], V.shape[3], 6, 6)
    for c0 in range(0, C, chunk):                                     # fp32 accumulation order ~ kernel's
        M += torch.einsum('bcxyil,kcil->bkxyil', V[:, c0:c0 + chunk], U[:, c0:c0 + chunk])
    Y = torch.einsum('ij,bkxyjl,ml->bkxyim', AT, M, AT)               # [B,K,th,tw,4,4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, H, W)


for (C, K, H) in ((64, 64, 32), (128, 128, 28), (256, 256, 28), (512, 512, 28)):
    x = torch.relu(torch.randn(2, C, H, H))                           # post-ReLU activations
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    d32 = F.conv2d(x, w, padding=1)
    y = wino43(x, w)
    sc = ref.abs().max()
    print(f'C={C} K={K}: direct fp32 err {((d32 - ref).abs().max() / sc).item():.2e}   F(4,3) fp32 err {((y - ref).abs().max() / sc).item():.2e}'
          f'   rms {((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e}')
