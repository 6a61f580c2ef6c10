"""The data-parallel path on the GPU box (one MI355X): RCCL itself at world size 1 on the real arena, and the
self-launching ``bench.py --gpus 2`` rehearsed with both ranks on the one device (gloo transport; RCCL refuses two ranks on
one GPU).  Each case runs in child processes so that process groups never leak into the test session."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    env.pop('WORLD_SIZE', None)
    env.update(extra)
    return env


def test_rccl_world_size_one_on_the_arena(hip_lib, tmp_path):
    """backend 'nccl' (= RCCL) initialises on this box and all-reduces the optimizer's flat gradient arena in place;
    FlatGradReducer + hip_scale + the fused clip+Adam step run on top of it."""
    script = tmp_path / 'rccl1.py'
    script.write_text(textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_PORT='29541')
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        from lct_vqa_amd import parallel
        from lct_vqa_amd.optim import FusedAdam
        from lct_vqa_amd._lib import lib
        lib.load()
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(33, 77), torch.nn.Tanh(), torch.nn.Linear(77, 5)).cuda()
        opt = FusedAdam(m.parameters(), lr=1e-3, flatten=True)
        red = parallel.FlatGradReducer(opt.flat['params'], opt.flat_grad, scale_fn=parallel.hip_scale)
        opt.zero_grad()
        m(torch.randn(8, 33, device='cuda')).sum().backward()
        before = opt.flat_grad.clone()
        h = dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM, async_op=True)      # what FlatGradReducer._launch issues
        h.wait()
        parallel.hip_scale(1.0, opt.flat_grad)
        red.finish()
        torch.cuda.synchronize()
        assert torch.equal(before, opt.flat_grad) and before.abs().sum() > 0
        p0 = opt.flat_param.clone()
        opt.step(max_grad_norm=5)
        assert not torch.equal(p0, opt.flat_param)
        dist.barrier()
        dist.destroy_process_group()
        print('RCCL_OK', torch.cuda.nccl.version())
    ''' % ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_launches_its_own_ranks(hip_lib):
    """``python bench.py --gpus 2`` with no launcher around it starts its two ranks (torch.distributed.run as a child) and
    prints ONE JSON line with n_gpus = 2 and the whole-job rate."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--batch', '32', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600,
                       env=_env(LCTVQA_REHEARSE_ONE_GPU='1'), cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 64 and out['scaling'] == 'weak'
    assert out['value'] > 0 and abs(out['value'] - 64 / (out['ms_per_step'] * 1e-3)) < 1e-2 * out['value']
    assert 0 < out['roofline']['frac'] <= 1.0
    # the line carries what lets a reader verify its ranks: transport, one device entry per rank gathered over the group,
    # bucket layout, all-reduces launched by the backward hooks, time finish() waited
    dp = out['dp']
    assert dp['world'] == 2 and dp['backend'] == 'gloo' and dp['rehearsal_one_gpu'] is True
    assert [d['rank'] for d in dp['devices']] == [0, 1] and len({d['pid'] for d in dp['devices']}) == 2
    assert dp['distinct_devices'] == 1                          # the rehearsal: both ranks on the one GPU, and the line says so
    assert dp['buckets'] == 3 and dp['steps'] == 2 and dp['launched_during_backward_per_step'] == 3.0
    assert dp['launched_by_finish_per_step'] == 0.0 and dp['allreduce_ms_per_step'] > 0.0


def test_rccl_failure_is_loud(hip_lib):
    """Two ranks asked to use RCCL on a box with ONE GPU (no rehearsal switch): the bench must exit non-zero naming the
    cause -- never fall back to another transport and print a line."""
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip('more than one GPU visible: RCCL would simply work')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--batch', '8', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert 'only 1 GPUs are visible' in r.stderr or 'did not come up' in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_sync_bn_dp2_equals_the_single_process_batch(hip_lib, tmp_path, dtype):
    """SyncBN (parallel.set_sync_bn; SURVEY.md section 8e BN caveat, reference BatchNorms pcdarts/operations.py:29,43,58,62,98): two ranks
    on half batches with the per-channel sums all-reduced inside the statistics finish == ONE process on the full batch -- trunk
    output, running statistics on both ranks, rank-averaged gradients (fp32 path: output and running statistics 1e-5, gradients 2e-5; bf16-native path: outputs up to a bf16
    rounding, gradient cosine 0.99: its activations are bf16 and a pooled sum rounds differently from a single pass).  Both
    ranks share the one GPU (gloo; RCCL refuses two ranks on a device)."""
    script = tmp_path / 'syncbn.py'
    script.write_text(textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        rank = int(os.environ['RANK'])
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=2)
        from lct_vqa_amd import parallel, ops
        from lct_vqa_amd.resnet import ResNet152Trunk
        from lct_vqa_amd._lib import lib
        lib.load()
        dtype = %r
        ops.set_compute_dtype(dtype)
        B, S = 8, 64
        g = torch.Generator().manual_seed(5)
        img = torch.randn(B, 3, S, S, generator=g).cuda()
        w = torch.randn(B, 2048, generator=g).cuda()
        def run(m, x, wt, denom):
            m.train()
            y = m(x)
            ((y * wt).sum() / denom).backward()
            return y.detach()
        torch.manual_seed(1)
        ref = ResNet152Trunk((1, 1, 1, 1)).cuda()
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data.uniform_(0.75, 1.25); mod.bias.data.normal_(0, 0.1)
        sd = {k: v.clone() for k, v in ref.state_dict().items()}
        parallel.set_sync_bn(False)
        y_full = run(ref, img, w, B)                                  # one process, whole batch
        m = ResNet152Trunk((1, 1, 1, 1)).cuda()
        m.load_state_dict(sd)
        parallel.set_sync_bn(True)
        lo, hi = rank * B // 2, (rank + 1) * B // 2
        y = run(m, img[lo:hi], w[lo:hi], B // 2)
        torch.cuda.synchronize()
        def rel(a, b): return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
        def cos(a, b): return (a.double().flatten() @ b.double().flatten() / (a.double().norm() * b.double().norm()).clamp_min(1e-30)).item()
        tol = 1e-5 if dtype == 'f32' else 1e-2
        assert rel(y, y_full[lo:hi]) < tol, ('output', rel(y, y_full[lo:hi]))
        for (n, b), (_, c) in zip(m.named_buffers(), ref.named_buffers()):
            if 'running' in n:
                assert rel(b, c) < (1e-5 if dtype == 'f32' else 1e-3), (n, rel(b, c))
            else:
                assert int(b) == int(c), n
        worst = 0.0
        for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            gavg = p.grad.clone()
            dist.all_reduce(gavg)
            gavg /= 2
            if dtype == 'f32':
                assert rel(gavg, q.grad) < 2e-5, (n, rel(gavg, q.grad))
            else:
                assert cos(gavg, q.grad) > 0.99, (n, cos(gavg, q.grad))
        # without SyncBN the same halves do NOT reproduce the full batch (the test would be vacuous otherwise)
        m2 = ResNet152Trunk((1, 1, 1, 1)).cuda()
        m2.load_state_dict(sd)
        parallel.set_sync_bn(False)
        y2 = run(m2, img[lo:hi], w[lo:hi], B // 2)
        assert rel(y2, y_full[lo:hi]) > 10 * tol
        dist.barrier()
        dist.destroy_process_group()
        print('SYNCBN_OK', rank)
    ''' % (ROOT, dtype)))
    port = '29547' if dtype == 'f32' else '29548'
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_env(RANK=str(r), WORLD_SIZE='2', MASTER_PORT=port)) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and 'SYNCBN_OK' in so, (so[-2000:], se[-4000:])


def test_sync_bn_dp2_pcdarts_search_network(hip_lib, tmp_path):
    """SyncBN for the PC-DARTS search encoder (reference BatchNorms pcdarts/operations.py:29,43,58,62,98; model_search.py:111-113): under
    parallel.set_sync_bn the network runs its per-primitive kernels (the fused edge kernels take their statistics inside and cannot
    all-reduce them) and every BatchNorm pools the per-channel sums of the ranks.  Two ranks on half batches == one process (fused
    edges, the default path) on the full batch: features 1e-4, running statistics 1e-4, rank-averaged gradients of every weight and of
    alpha / beta within 2e-3 of their largest entry (a ReLU whose pre-activation differs in the last bits between the two summation orders
    flips: the architects' parity notes).  Without SyncBN the halves do not reproduce the full batch."""
    script = tmp_path / 'syncbn_darts.py'
    script.write_text(textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        rank = int(os.environ['RANK'])
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=2)
        from lct_vqa_amd import parallel, darts_ops
        from lct_vqa_amd.basic_vqa import config
        config.DEVICE = torch.device('cuda')
        from lct_vqa_amd.basic_vqa.pcdarts.model_search import Network
        from lct_vqa_amd._lib import lib
        lib.load()
        B, S = 8, 32
        g = torch.Generator().manual_seed(7)
        img = torch.randn(B, 3, S, S, generator=g).cuda()
        w = torch.randn(B, 256 * 7 * 7, generator=g).cuda()
        def run(m, x, wt, denom):
            m.train()
            for a in m.arch_parameters():
                a.grad = None
            y = m(x)
            ((y * wt).sum() / denom).backward()
            return y.detach()
        torch.manual_seed(3)
        ref = Network(16, 512, 3).cuda()
        for a in ref.arch_parameters():
            a.data.normal_(0, 0.5)
        sd = {k: v.clone() for k, v in ref.state_dict().items()}
        arch = [a.detach().clone() for a in ref.arch_parameters()]
        parallel.set_sync_bn(False)
        assert darts_ops.fused_edge_active()
        y_full = run(ref, img, w, B)                                  # one process, whole batch, fused edges
        def twin():
            m = Network(16, 512, 3).cuda()
            m.load_state_dict(sd)
            for a, r in zip(m.arch_parameters(), arch):
                a.data.copy_(r)
            return m
        m = twin()
        parallel.set_sync_bn(True)
        assert not darts_ops.fused_edge_active() and darts_ops.sync_bn_active()
        lo, hi = rank * B // 2, (rank + 1) * B // 2
        y = run(m, img[lo:hi], w[lo:hi], B // 2)
        torch.cuda.synchronize()
        def rel(a, b): return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
        assert rel(y, y_full[lo:hi]) < 1e-4, ('output', rel(y, y_full[lo:hi]))
        for (n, b), (_, c) in zip(m.named_buffers(), ref.named_buffers()):
            if 'running' in n:
                assert rel(b, c) < 1e-4, (n, rel(b, c))
            else:
                assert int(b) == int(c), n
        pairs = list(zip(m.named_parameters(), ref.named_parameters())) + [((f'arch{i}', a), (None, r)) for i, (a, r) in enumerate(zip(m.arch_parameters(), ref.arch_parameters()))]
        for (n, p), (_, q) in pairs:
            gavg = p.grad.clone()
            dist.all_reduce(gavg)
            gavg /= 2
            assert rel(gavg, q.grad) < 2e-3, (n, rel(gavg, q.grad))
        m2 = twin()
        parallel.set_sync_bn(False)
        y2 = run(m2, img[lo:hi], w[lo:hi], B // 2)
        assert rel(y2, y_full[lo:hi]) > 1e-2                          # per-rank statistics: not the full batch
        dist.barrier()
        dist.destroy_process_group()
        print('SYNCBN_DARTS_OK', rank)
    ''' % (ROOT,)))
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_env(RANK=str(r), WORLD_SIZE='2', MASTER_PORT='29549')) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and 'SYNCBN_DARTS_OK' in so, (so[-2000:], se[-4000:])


def test_trainable_bf16_resnet_under_dp_with_the_gradient_sink(hip_lib):
    """BASELINE configs[2] (ResNet-152 + GRU, bf16, encoder trainable, data parallel) rehearsed on two ranks sharing the one GPU: with
    ops.grad_sink the bottleneck nodes accumulate their parameter gradients into the optimizer arena themselves and return None, and
    the reducer's bucket counters must still see every parameter exactly once per backward pass (torch runs a leaf's
    post-accumulate-grad hooks for a None gradient as well; ops.sink_done fires them itself only where it does not): every bucket is
    launched by the hooks during backward, none by finish(), and no hook fires twice (FlatGradReducer raises on that)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'resnet_gru', '--train-encoder', '--dtype',
                        'bf16', '--batch', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True,
                       timeout=600, env=_env(LCTVQA_REHEARSE_ONE_GPU='1'), cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(lines[0])
    dp = out['dp']
    assert out['n_gpus'] == 2 and dp['world'] == 2 and dp['buckets'] >= 8
    assert dp['launched_during_backward_per_step'] == float(dp['buckets']) and dp['launched_by_finish_per_step'] == 0.0
    assert 5.0 < out['loss'] < 10.0
