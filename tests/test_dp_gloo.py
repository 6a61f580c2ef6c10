"""CPU, world_size 2, gloo: the data-parallel reducer (lct_vqa_amd/parallel.py) must reproduce the
single-process large-batch gradient: mean over ranks of per-rank mean-loss gradients == gradient of the
mean loss over the concatenated batch (samples are independent in every layer of the path)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(3)
    return nn.Sequential(nn.Linear(20, 300), nn.Tanh(), nn.Linear(300, 17000), nn.Tanh(), nn.Linear(17000, 7))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lct_vqa_amd import parallel
    m = _model()
    params = [p for p in m.parameters()]
    flat = parallel.make_flat_grad(params)
    red = parallel.FlatGradReducer(params, flat, bucket_bytes=1 << 20)     # several buckets
    assert len(red.buckets) >= 2
    torch.manual_seed(11)
    x = torch.randn(8, 20)
    y = torch.randint(7, (8,))
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for it in range(2):                                                   # hooks must re-arm every step
        flat.zero_()
        loss = nn.functional.cross_entropy(m(xs), ys)
        loss.backward()
        red.finish()
    if rank == 0:
        torch.save(flat.clone(), out)
    dist.destroy_process_group()


def test_dp2_matches_single_process(tmp_path):
    out = str(tmp_path / 'g.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m = _model()
    torch.manual_seed(11)
    x = torch.randn(8, 20)
    y = torch.randint(7, (8,))
    nn.functional.cross_entropy(m(x), y).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-7)


def _arena_worker(rank, world, port, out):
    """The objects bench.py uses: optim.FusedAdam(flatten=True)'s padded arena + FlatGradReducer on its flat gradient (CPU
    tensors here, so only the optimizer STEP -- a HIP kernel -- is left out)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lct_vqa_amd import parallel
    from lct_vqa_amd.optim import FusedAdam
    m = _model()
    opt = FusedAdam(m.parameters(), lr=1e-3, flatten=True)
    assert all(o % 4 == 0 for o in opt.flat['offsets']) and opt.flat_grad.numel() % 4 == 0
    assert opt.flat_grad.numel() > sum(p.numel() for p in m.parameters())          # 17000*300, 7: padding exists
    red = parallel.FlatGradReducer(opt.flat['params'], opt.flat_grad, bucket_bytes=1 << 20)
    assert len(red.buckets) >= 2 and red.buckets[0][0] == 0 and red.buckets[-1][1] == opt.flat_grad.numel()
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))           # the buckets tile the arena
    torch.manual_seed(11)
    x = torch.randn(8, 20)
    y = torch.randint(7, (8,))
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for it in range(2):
        opt.zero_grad()
        nn.functional.cross_entropy(m(xs), ys).backward()
        red.finish()
    assert opt.flat['touched'] == set(range(len(opt.flat['params'])))
    if rank == 0:
        torch.save([p.grad.clone() for p in m.parameters()], out)
    dist.destroy_process_group()


def test_dp2_on_the_optimizer_arena(tmp_path):
    out = str(tmp_path / 'g.pt')
    mp.spawn(_arena_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m = _model()
    torch.manual_seed(11)
    x = torch.randn(8, 20)
    y = torch.randint(7, (8,))
    nn.functional.cross_entropy(m(x), y).backward()
    for g, p in zip(got, m.parameters()):
        assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-7)


def test_arena_skips_parameters_backward_did_not_reach():
    """torch.optim.Adam leaves grad-None parameters alone; the arena records which parameters backward reached."""
    from lct_vqa_amd.optim import FusedAdam
    a, b, c = nn.Linear(3, 5), nn.Linear(5, 7), nn.Linear(7, 2)
    opt = FusedAdam(list(a.parameters()) + list(b.parameters()) + list(c.parameters()), flatten=True)
    opt.zero_grad()
    c(torch.randn(2, 7)).sum().backward()                      # only the last module
    assert opt.flat['touched'] == {4, 5}
    assert opt._runs() == [[4, 5, 1]]
    opt.flat['touched'].clear()
    a(torch.randn(2, 3)).sum().backward()
    c(torch.randn(2, 7)).sum().backward()
    assert opt._runs() == [[0, 1, 1], [4, 5, 2]]             # per-parameter step counts, like torch's state['step']
    assert float(opt.state[c.weight]['step']) == 2.0 and float(opt.state[b.weight]['step']) == 0.0


def test_single_process_reducer_is_noop():
    from lct_vqa_amd import parallel
    m = _model()
    params = list(m.parameters())
    flat = parallel.make_flat_grad(params)
    red = parallel.FlatGradReducer(params, flat)
    nn.functional.cross_entropy(m(torch.randn(4, 20)), torch.randint(7, (4,))).backward()
    before = flat.clone()
    red.finish()
    assert torch.equal(before, flat) and flat.abs().sum() > 0


def _architect_worker(rank, world, port, out):
    """A whole 2-level architecture step under data parallelism: each rank sees half of the train and of the valid batch,
    bilevel.data_parallel() averages every gradient the step takes."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lct_vqa_amd import bilevel
    from test_bilevel import Toy, _data
    torch.manual_seed(5)
    m = Toy()
    x, y = _data(6, n=32)
    vx, vy = _data(7, n=32)
    sl = slice(rank * 16, (rank + 1) * 16)
    bilevel.data_parallel(True)
    try:
        w = list(m.parameters())
        ahead = bilevel.virtual_step(m, bilevel.grads(m._loss(x[sl], y[sl]), w, 1), 0.05)
        g = bilevel.grads(ahead._loss(vx[sl], vy[sl]), ahead.arch_parameters() + list(ahead.parameters()), 1)
        implicit = bilevel.fd_grad(w, g[1:], lambda: bilevel.grads(m._loss(x[sl], y[sl]), m.arch_parameters()))
        dalpha = g[0] - 0.05 * implicit[0]
    finally:
        bilevel.data_parallel(False)
    torch.save(dalpha.clone(), out + str(rank))
    dist.destroy_process_group()


def test_architecture_step_under_dp_equals_the_large_batch_step(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from lct_vqa_amd import bilevel
    from test_bilevel import Toy, _data
    out = str(tmp_path / 'a')
    mp.spawn(_architect_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a0, a1 = torch.load(out + '0'), torch.load(out + '1')
    assert torch.equal(a0, a1)                                  # every rank holds the same architecture gradient
    torch.manual_seed(5)
    m = Toy()
    x, y = _data(6, n=32)
    vx, vy = _data(7, n=32)
    w = list(m.parameters())
    ahead = bilevel.virtual_step(m, bilevel.grads(m._loss(x, y), w, 1), 0.05)
    g = bilevel.grads(ahead._loss(vx, vy), ahead.arch_parameters() + list(ahead.parameters()), 1)
    implicit = bilevel.fd_grad(w, g[1:], lambda: bilevel.grads(m._loss(x, y), m.arch_parameters()))
    ref = g[0] - 0.05 * implicit[0]
    assert torch.allclose(a0, ref, rtol=1e-9, atol=1e-12)


def _w_arena_worker(rank, world, port, out):
    """The W model's REAL trainable set (BASELINE configs[1]: 13 tensors, 21.05 M floats = 84 MB -> three 32 MiB buckets) in
    FusedAdam's arena: the backward hooks must launch every bucket's all-reduce BEFORE finish() is called (that is the
    overlap with backward), finish() only waits, and the averaged gradients are the large-batch ones."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lct_vqa_amd import parallel
    from lct_vqa_amd.optim import FusedAdam
    shapes = _w_trainable_shapes()
    torch.manual_seed(21)
    ps = [nn.Parameter(torch.randn(s) * 0.01) for s in shapes]
    opt = FusedAdam(ps, lr=1e-3, flatten=True)
    assert opt.flat_grad.numel() >= 21_045_883
    red = parallel.FlatGradReducer(opt.flat['params'], opt.flat_grad).timing(True)
    assert len(red.buckets) == 3
    torch.manual_seed(100 + rank)
    coef = [torch.randn(()) for _ in ps]
    for it in range(2):
        opt.zero_grad()
        loss = sum((c * p).sum() + 0.5 * (rank + 1) * p.square().sum() for c, p in zip(coef, ps))
        loss.backward()
        assert all(b[2] == 0 for b in red.buckets) and len(red.handles) == 3       # all launched by the hooks, none left
        red.finish()
    d = red.describe()
    assert d['steps'] == 2 and d['launched_during_backward_per_step'] == 3.0 and d['launched_by_finish_per_step'] == 0.0
    assert d['buckets'] == 3 and sum(d['bucket_bytes']) == opt.flat_grad.numel() * 4 and d['allreduce_wait_ms_per_step'] >= 0.0
    if rank == 0:
        torch.save([[c for c in coef], [p.grad[..., :3].clone() for p in ps]], out)
    else:
        torch.save([c for c in coef], out + '.r1')
    dist.destroy_process_group()


def _w_trainable_shapes():
    E, H, We, A, QV = 512, 512, 300, 3129, 17858
    return [(E, 4096), (E,), (QV, We), (4 * H, We), (4 * H, H), (4 * H,), (4 * H,), (E, 2 * H), (E,), (A, E), (A,), (A, A), (A,)]


def test_dp2_w_model_arena_three_buckets_launched_by_the_hooks(tmp_path):
    assert sum(torch.Size(s).numel() for s in _w_trainable_shapes()) == 21_045_883     # = models.VqaModel(**bench.DIMS).trainable_parameters()
    out = str(tmp_path / 'w.pt')
    mp.spawn(_w_arena_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    coef0, grads = torch.load(out)
    coef1 = torch.load(out + '.r1')
    torch.manual_seed(21)
    ps = [torch.randn(s) * 0.01 for s in _w_trainable_shapes()]
    for p, g, c0, c1 in zip(ps, grads, coef0, coef1):
        want = 0.5 * ((c0 + p) + (c1 + 2.0 * p))                  # mean over the two ranks of d loss_r / d p
        assert torch.allclose(g, want[..., :3], rtol=1e-5, atol=1e-7)
