#!/usr/bin/env python3
"""Dump every output of one fused-edge case (tests/test_gpu_edge._case inputs) for the library in LCTVQA_LIB and compare with
the CPU oracle entry by entry: python tools/edge_debug.py C stride B H W [seed]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _util import rel_err
C, stride, B, H, W = map(int, sys.argv[1:6])
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 100 + C + stride
from lct_vqa_amd._lib import lib
lib.load()
from lct_vqa_amd.basic_vqa import config
from lct_vqa_amd.basic_vqa.pcdarts import model_search as MS
from oracle import darts_oracle as DO
config.DEVICE = torch.device('cuda')
g = torch.Generator().manual_seed(seed)
torch.manual_seed(seed)
op = MS.MixedOp(C, stride)
for m in op.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.3, generator=g); m.running_var.uniform_(0.5, 1.5, generator=g)
P = {'e.' + k: v.detach().clone() for k, v in op.state_dict().items()}
names = [n for n, _ in op.named_parameters()]
x = torch.randn(B, C, H, W, generator=g); w = torch.softmax(torch.randn(8, generator=g), 0); beta = torch.rand(1, generator=g) + 0.5
Ho, Wo = H // stride, W // stride
prev = torch.randn(B, C, Ho, Wo, generator=g); gout = torch.randn(B, C, Ho, Wo, generator=g)
xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), beta.clone().requires_grad_()
Pg = {k: (v.clone().requires_grad_(True) if k[2:] in names else v) for k, v in P.items()}
pr = prev.clone().requires_grad_()
ref = pr + br * DO._mixed_op(xr, wr, Pg, 'e', stride, True, {})
gref = torch.autograd.grad(ref, [xr, wr, br, pr] + [Pg['e.' + n] for n in names], gout)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
mod = op.cuda().train()
xd, wd, bd, pd = nhwc(x).requires_grad_(), w.cuda().requires_grad_(), beta.cuda().requires_grad_(), nhwc(prev).requires_grad_()
y = mod(xd, wd, pd * 1.0, bd)
gh = torch.autograd.grad(y, [xd, wd, bd, pd] + [p for _, p in mod.named_parameters()], nhwc(gout))
print('lib', os.environ.get('LCTVQA_LIB'), 'out', rel_err(y.permute(0, 3, 1, 2).cpu(), ref))
for lab, a, r in zip(['dx', 'dalpha', 'dbeta', 'dprev'] + names, gh, gref):
    a = a.permute(0, 3, 1, 2).cpu() if a.dim() == 4 and lab in ('dx', 'dprev') else a.cpu()
    e = rel_err(a.reshape(r.shape), r)
    extra = ''
    if lab == 'dx':
        d = (a - r).abs()
        per_c = d.amax(dim=(0, 2, 3))
        extra = f' per-channel max err {[round(float(v), 5) for v in per_c[:C // 4]]} ref max {float(r.abs().max()):.4f}; worst at {tuple(int(i) for i in torch.nonzero(d == d.max())[0])}'
    print(f'  {lab:28s} {e:.3e}{extra}')
torch.save([t.detach().cpu() for t in gh], f'/tmp/edge_dbg_{os.path.basename(os.environ.get("LCTVQA_LIB", "default"))}.pt')
