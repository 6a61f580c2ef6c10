#!/usr/bin/env python3
"""Which parts of the PC-DARTS path survive HIP-graph capture on this ROCm?  Runs one scenario per child process
(a capture that crashes the runtime takes only its own process down) and prints a verdict per scenario.
    python tools/graph_probe.py            # all scenarios
    python tools/graph_probe.py <name>     # one scenario, in this process"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def capture(fn, replays=3):
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / replays * 1e3


def scenario(name):
    import torch
    import torch.nn as nn
    from lct_vqa_amd.basic_vqa import config
    config.DEVICE = torch.device('cuda')
    from lct_vqa_amd import darts_ops as D
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network, MixedOp
    torch.manual_seed(0)
    dev = 'cuda'
    x = torch.randn(4, 32, 32, 16, device=dev, requires_grad=True)
    if name == 'torch_only':
        w = torch.randn(16, 16, device=dev, requires_grad=True)
        fn = lambda: torch.autograd.grad((x @ w).sum(), [x, w])
    elif name == 'dwconv_fwd':
        w = torch.randn(16, 3, 3, device=dev)
        fn = lambda: D.DwConvFn.apply(x.detach(), w, 1, 1, 1, True)
    elif name == 'dwconv_fwd_bwd':
        w = torch.randn(16, 3, 3, device=dev, requires_grad=True)
        fn = lambda: torch.autograd.grad(D.DwConvFn.apply(x, w, 1, 1, 1, True).sum(), [x, w])
    elif name == 'pwconv_fwd_bwd':
        w = torch.randn(16, 16, 1, 1, device=dev, requires_grad=True)
        fn = lambda: torch.autograd.grad(D.PwConvFn.apply(x, w, True).sum(), [x, w])
    elif name == 'bn_fwd_bwd':
        bn = nn.BatchNorm2d(16, affine=False).to(dev).train()
        fn = lambda: torch.autograd.grad(D.batchnorm(bn, x).sum(), [x])
    elif name == 'pool_fwd_bwd':
        fn = lambda: torch.autograd.grad(D.Pool3x3Fn.apply(x, 1, 0).sum(), [x])
    elif name == 'mixedop_fwd':
        op = MixedOp(16, 1).to(dev).train()
        wts = torch.softmax(torch.randn(8, device=dev), 0)
        fn = lambda: op(x.detach(), wts)
    elif name == 'mixedop_fwd_bwd':
        op = MixedOp(16, 1).to(dev).train()
        wts = torch.softmax(torch.randn(8, device=dev), 0).requires_grad_()
        fn = lambda: torch.autograd.grad(op(x, wts).sum(), [x, wts] + list(op.parameters()))
    elif name in ('network_fwd', 'network_fwd_bwd'):
        net = Network(16, 512, 4).to(dev).train()
        img = torch.randn(2, 3, 64, 64, device=dev)
        params = list(net.parameters()) + net.arch_parameters()
        if name == 'network_fwd':
            def fn():
                with torch.no_grad():
                    return net(img)
        else:
            fn = lambda: torch.autograd.grad(net(img).sum(), params)
    else:
        raise SystemExit(f'unknown scenario {name}')
    _, ms = capture(fn)
    print(f'{name}: capture + replay OK, {ms:.2f} ms per replay', flush=True)


NAMES = ['torch_only', 'dwconv_fwd', 'dwconv_fwd_bwd', 'pwconv_fwd_bwd', 'bn_fwd_bwd', 'pool_fwd_bwd', 'mixedop_fwd', 'mixedop_fwd_bwd',
         'network_fwd', 'network_fwd_bwd']

if __name__ == '__main__':
    if len(sys.argv) > 1:
        scenario(sys.argv[1])
    else:
        for n in NAMES:
            r = subprocess.run([sys.executable, '-X', 'faulthandler', os.path.abspath(__file__), n], capture_output=True, text=True, timeout=300)
            ok = [line for line in r.stdout.splitlines() if 'capture + replay OK' in line]
            if r.returncode == 0 and ok:
                print(ok[0], flush=True)
            else:
                tail = [line for line in (r.stderr or '').splitlines() if line.strip() and 'Warning' not in line and 'amdgpu.ids' not in line][-6:]
                print(f'{n}: FAILED rc={r.returncode}  ' + ' | '.join(tail), flush=True)
