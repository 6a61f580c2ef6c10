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
