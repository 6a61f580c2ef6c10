#!/usr/bin/env python3
"""Time one forward+backward of the PC-DARTS encoder (Network(16,512,4)) and of a darts_vqa train step.
    python tools/bench_darts.py [--batch 16] [--size 224]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--graph', action='store_true', help='also capture the encoder forward+backward into one HIP graph and time replays')
    ap.add_argument('--vqa', action='store_true', help='also time a darts_vqa weight step and an Architect step')
    ap.add_argument('--lct', action='store_true', help='also time the three stages of a basic_vqa LCT search iteration (ArchitectLct.step, stage 1, stage 2)')
    args = ap.parse_args()
    from lct_vqa_amd.basic_vqa import config
    config.DEVICE = torch.device('cuda')
    from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
    torch.manual_seed(0)
    net = Network(16, 512, 4).to('cuda').train()
    x = torch.randn(args.batch, 3, args.size, args.size, device='cuda')
    for it in range(args.iters + 1):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = net(x)
        g = torch.autograd.grad(out.sum(), list(net.parameters()) + net.arch_parameters())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    macs = 1408.9e6 * (args.size / 224.0) ** 2
    print(f'DARTS encoder fwd+bwd B={args.batch} {args.size}x{args.size}: {dt * 1e3:.1f} ms  '
          f'({args.batch / dt:.1f} img/s, ~{3 * 2 * macs * args.batch / dt / 1e12:.2f} TFLOP/s, peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB)')
    if args.graph:
        # the same forward+backward captured once into a HIP graph and replayed: no host work per launch
        del out, g
        params = list(net.parameters()) + net.arch_parameters()
        step = lambda: torch.autograd.grad(net(x).sum(), params)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ggrads = step()
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in ggrads]
        ref = step()
        err = max(((a - b).abs().max() / (b.abs().max() + 1e-30)).item() for a, b in zip(got, ref))
        t0 = time.perf_counter()
        for _ in range(args.iters):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        print(f'  as one HIP graph: {dt * 1e3:.1f} ms per replay ({args.batch / dt:.1f} img/s), gradients vs eager max rel diff {err:.1e}')
    if args.vqa:
        if not args.graph:
            del out, g
        del net
        darts_vqa_step(args)
    if args.lct:
        lct_search_iteration(args)


def darts_vqa_step(args):
    """BASELINE configs[4] in fp32 on one GPU: darts_vqa VqaModel(img_encoder_type='darts') weight step
    (CE(ans) + CE(qst), clip, Adam) and the 2-level Architect.step that the driver runs beside it."""
    import types
    from lct_vqa_amd.darts_vqa import config as dconfig, factory
    from lct_vqa_amd.basic_vqa import config as bconfig
    dconfig.DEVICE = bconfig.DEVICE = torch.device('cuda')
    a = types.SimpleNamespace(unified=False, embed_size=512, word_embed_size=300, num_layers=1, hidden_size=512, arch_type='darts',
                              learn_rate=1e-3, step_size=10, gamma=0.1, arch_learn_rate=6e-4, arch_wt_decay=1e-3, qst_only=False)
    ds = types.SimpleNamespace(qst_vocab=types.SimpleNamespace(vocab_size=17858), ans_vocab=types.SimpleNamespace(vocab_size=3129))
    torch.manual_seed(10)
    model = factory.get_vqa_model(a, ds).to('cuda').train()
    opt = factory.get_optimizer(a, model)
    arch = factory.get_architect(a, model)
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(10)
    img = torch.randn(B, 3, S, S, generator=g).cuda()
    qst = torch.randint(17858, (B, 14), generator=g); qst[:, 0] = 2
    qst = qst.cuda()
    lab = torch.randint(3129, (B,), generator=g).cuda()

    def weight_step():
        opt.zero_grad()
        loss = model._loss(img, qst, lab, False)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()

    for name, fn in (('weight step', weight_step), ('Architect.step (2nd order)', lambda: arch.step(img, qst, lab, img, qst, lab, 1e-3, opt))):
        for it in range(args.iters + 1):
            if it == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        print(f'darts_vqa {name} B={B} {S}x{S}: {dt * 1e3:.1f} ms  ({B / dt:.1f} pairs/s)')


def lct_search_iteration(args):
    """BASELINE configs[3] with the search encoder, fp32, one GPU: the three stages of experiment.py:199-257 through the factories --
    ArchitectLct.step (3-level: unrolled examiner and W, generated tests, finite-difference Hessian-vector products), stage 1
    (examiner step), stage 2 (generate + W soft-loss step)."""
    import types
    from lct_vqa_amd.basic_vqa import architect_factory, config, model_factory
    from lct_vqa_amd import ops
    config.DEVICE = torch.device('cuda')
    config.ARCH_TYPE = 'darts'
    config.SKIP_STAGE2 = False
    ds = types.SimpleNamespace(qst_vocab=types.SimpleNamespace(vocab_size=17858), ans_vocab=types.SimpleNamespace(vocab_size=3129))
    torch.manual_seed(10)
    ef = model_factory.get_ef_model(ds).to('cuda').train()
    w = model_factory.get_w_model(ds).to('cuda').train()
    ef.qst_encoder.max_length = 14
    ef_opt, w_opt = model_factory.get_ef_optimizer(ef), model_factory.get_w_optimizer(w)
    arch = architect_factory.get_architect(ef, w, ef_opt, w_opt)
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(10)
    img = torch.randn(B, 3, S, S, generator=g).cuda()
    qst = torch.randint(17858, (B, 14), generator=g); qst[:, 0] = 2
    qst = qst.cuda()
    lab = torch.randint(3129, (B,), generator=g).cuda()

    def stage1():
        ef_opt.zero_grad()
        ef._loss(img, qst, lab).backward()
        torch.nn.utils.clip_grad_norm_(ef.parameters(), config.GRAD_CLIP)
        ef_opt.step()

    def stage2():
        with torch.no_grad():
            pq, pa = ef.generate(img)
            pa = ops.softmax_rows(pa, 1.0)
        w_opt.zero_grad()
        w._soft_loss(img, qst, lab, pq, pa).backward()
        torch.nn.utils.clip_grad_norm_(w.parameters(), config.GRAD_CLIP)
        w_opt.step()

    stage3 = lambda: arch.step(img, qst, lab, img, qst, lab, config.LEARNING_RATE, config.LEARNING_RATE)
    for name, fn in (('stage 3 ArchitectLct.step', stage3), ('stage 1 examiner step', stage1), ('stage 2 generate + W step', stage2)):
        for it in range(args.iters + 1):
            if it == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        print(f'LCT search {name} B={B} {S}x{S}: {dt * 1e3:.1f} ms  ({B / dt:.1f} pairs/s)')


if __name__ == '__main__':
    main()
