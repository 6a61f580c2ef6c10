"""Stage-by-stage comparison of the bf16 bottleneck backward with float64 (debug helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lct_vqa_amd import resnet_bf16 as R, ops
from lct_vqa_amd._lib import lib
lib.load()
DEV = 'cuda'
BF = torch.bfloat16
torch.manual_seed(0)
B, H, W, Cin, planes = 3, 56, 56, 256, 64
P = B * H * W
g = torch.Generator().manual_seed(1)
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()
x = torch.randn(P, Cin, generator=g).to(BF).to(DEV)
w1 = (torch.randn(planes, Cin, 1, 1, generator=g) * 0.1).to(DEV)
w1b, w1t = R._w1x1_bf16(w1)
print('w1b', rel(w1b, w1.view(planes, Cin).to(BF)), 'w1t', rel(w1t, w1.view(planes, Cin).t().to(BF)))
bn = torch.nn.BatchNorm2d(planes).to(DEV).train()
bn.weight.data.uniform_(0.75, 1.25); bn.bias.data.normal_(0, 0.2)
# forward conv1 + bn1 + relu
z1, st, n = R.conv1x1(x, w1b)
zr = x.double() @ w1b.double().t()
print('z1', rel(z1, zr))
y1, c1, tr = R.bn_forward(z1, st, n, bn, None, True)
m, v = zr.mean(0), zr.var(0, unbiased=False)
print('mean', rel(c1[0], m), 'invstd', rel(c1[1], 1 / torch.sqrt(v + bn.eps)))
yr = F.relu((zr - m) / torch.sqrt(v + bn.eps) * bn.weight.double() + bn.bias.double())
print('y1', rel(y1, yr))
# backward of bn1 given dy1 with mask y1 (no g_out)
dy = torch.randn(P, planes, generator=g).to(BF).to(DEV)
zz = zr.clone().requires_grad_(True)
wg = bn.weight.detach().double().requires_grad_(True)
bg = bn.bias.detach().double().requires_grad_(True)
yy = F.relu(F.batch_norm(zz, None, None, wg, bg, True, 0.1, bn.eps))
yy.backward(dy.double())
dz1, dg1, db1 = R.bn_backward(dy, y1, z1, c1, bn.weight, tr)
print('dz1', rel(dz1, zz.grad), 'dg', rel(dg1, wg.grad), 'db', rel(db1, bg.grad))
# dgrad gemm with addend
add = torch.randn(P, Cin, generator=g).to(BF).to(DEV)
dx, _, _ = R.conv1x1(dz1, w1t, stats=False, addend=add)
print('dx', rel(dx, dz1.double() @ w1b.double() + add.double()))
# masked dgrad
msk = torch.randn(P, Cin, generator=g).to(BF).to(DEV)
dx2, _, _ = R.conv1x1(dz1, w1t, stats=False, mask=msk)
r2 = dz1.double() @ w1b.double()
print('dx masked', rel(dx2, torch.where(msk.double() > 0, r2, torch.zeros_like(r2))))
dw = R.wgrad(dz1, x, planes, Cin)
print('dw1', rel(dw.view(planes, Cin), dz1.double().t() @ x.double()))
# tiled 3x3 both ways + wgrad
w2 = (torch.randn(planes, planes, 3, 3, generator=g) * 0.05).to(DEV)
c = R._Conv3x3(w2, B, H, W)
print('tiled', c.tiled)
z2, st, n = c.forward(y1.view(B, H, W, planes))
z2r = F.conv2d(y1.double().view(B, H, W, planes).permute(0, 3, 1, 2), w2.to(BF).double(), padding=1).permute(0, 2, 3, 1)
print('z2', rel(z2, z2r))
dz2 = torch.randn(B, H, W, planes, generator=g).to(BF).to(DEV)
d1, masked = c.dgrad(dz2, y1.view(B, H, W, planes))
d1r = F.conv_transpose2d(dz2.double().permute(0, 3, 1, 2), w2.to(BF).double(), padding=1).permute(0, 2, 3, 1)
print('dy1', masked, rel(d1, d1r))
dw2 = R.wgrad(dz2.view(P, planes), y1, planes, planes, 9, H, W)
wref = torch.zeros(planes, planes, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
F.conv2d(y1.double().view(B, H, W, planes).permute(0, 3, 1, 2), wref, padding=1).backward(dz2.double().permute(0, 3, 1, 2))
print('dw2', rel(dw2, wref.grad))
