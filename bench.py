#!/usr/bin/env python3
"""Headline benchmark: image-question pairs/s of one train step of the W model
(basic_vqa VGG19+LSTM, fp32, 224x224 images, 14-token questions, 3129 answers, QV=17858)
= BASELINE.json configs[1], on N MI355X GPUs of one node (one process per GPU, RCCL all-reduce).

A step = forward (frozen VGG19 trunk under no_grad, dropout live) + CE loss + backward through the
trainable head + all-reduce (N>1) + global-norm clip(5) + Adam, on synthetic inputs already
resident in HBM.  Contract: W untimed warm-up steps, then exactly K timed steps bracketed by
barrier + synchronize; MAX over ranks; rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(``torch.distributed.run`` as a child process, before this process has touched the GPU) and relays rank 0's JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it on this driver

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                  # same guide: HBM3E spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0         # same guide: dense bf16 MFMA (AMD's 5 PF figure includes 2:1 sparsity)
DIMS = dict(embed_size=512, qst_vocab_size=17858, ans_vocab_size=3129, word_embed_size=300, num_layers=1, hidden_size=512)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (weak scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=6)
    ap.add_argument('--layer-report', default='', help='write the per-conv-layer timing table to this file')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='run trunk and head of a step back to back on one stream (default: the frozen trunk of batch i+1 '
                         'overlaps the head/backward/optimizer of batch i on a second HIP stream)')
    ap.add_argument('--head-priority', type=int, default=1,
                    help='pipelined mode: run the head (LSTM, fusion, classifier, backward, optimizer) on a HIGH-priority HIP stream, so that '
                         'its small kernels get the CU slots the trunk\'s convolution workgroups free up first (measured, same device: LCT bf16 '
                         '25.7 -> 24.2 ms/step; W fp32 / bf16 and ResNet+GRU unchanged within 0.3 %%); 0 = the default stream')
    ap.add_argument('--trunk-split', type=int, default=1,
                    help='run the frozen trunk of a batch as this many sub-batches on streams of their own (pipelined mode).  2 is '
                         '1.4 %% faster (tails of one part filled by the other); default 1 because overlapping conv kernels make the '
                         'per-launch durations behind the roofline fields meaningless')
    ap.add_argument('--head-after', type=int, default=None,
                    help='pipelined mode: hold the head of batch i until the trunk of batch i+1 has launched past conv layer N '
                         '(1..16; 0 = start it as soon as its own trunk is done).  Default 8 = after conv3_4 (fp32): the head shares the chip with the '
                         '28x28 layers; 0.8 %% faster than 0, 12 and later leave it too little time; 4 = after conv2_2 with the bf16 trunk '
                         '(11.78 vs 11.90 ms/step)')
    ap.add_argument('--classifier-with-head', action='store_true',
                    help='pipelined mode: the trunk stream runs vgg19.features + avgpool only; the classifier (fc6, fc7 + dropouts) runs with '
                         'the head on the main stream (same work per step, other stream)')
    ap.add_argument('--stem-after', type=int, default=0,
                    help='pipelined mode: run conv1_1 of batch i+1 on a stream of its own, released when the trunk of batch i has '
                         'launched past conv layer N (0 = conv1_1 stays at the head of its own trunk)')
    ap.add_argument('--workload', default='w', choices=['w', 'lct', 'resnet_gru', 'darts_vqa'],
                    help="'w' = BASELINE configs[1] (default, the headline metric); 'lct' = stage-1 + stage-2 step of "
                         "experiment.py:216-268 (EF fixed encoder + generate + W soft loss), reported under its own metric name; "
                         "'resnet_gru' = BASELINE configs[2]'s model shape (frozen ResNet-152 trunk + GRU + fusion head) train step, own metric name; "
                         "'darts_vqa' = BASELINE configs[4] in fp32 on one GPU: darts_vqa VqaModel(img_encoder_type='darts') weight step "
                         "(darts_vqa/experiment.py:188-200), own metric name; one 2nd-order Architect.step is timed beside it (not in value)")
    ap.add_argument('--lct-no-share', dest='lct_share_trunk', action='store_false',
                    help="--workload lct: evaluate the examiner's frozen trunk separately for stage 1 and stage 2 (default: once per batch)")
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="arithmetic of the frozen VGG19 trunk: 'f32' = the reference's (the headline); 'bf16' = bf16 activations/weights with fp32 "
                         "accumulation (BASELINE.json's bf16 configurations), reported under its OWN metric name, never as the headline")
    ap.add_argument('--no-grad-sink', action='store_true', help='A/B: let autograd accumulate every parameter gradient with an ATen add_ (ops.grad_sink off)')
    ap.add_argument('--train-encoder', action='store_true',
                    help="--workload resnet_gru: train the ResNet-152 trunk too (batch-statistics BatchNorm, full backward) instead of the frozen trunk")
    ap.add_argument('--conv-algo', default=None, choices=['winograd', 'winograd2', 'direct'],
                    help='trunk convolution algorithm (default: LCTVQA_CONV_ALGO or winograd = F(4x4,3x3) / F(2x2,3x3) / direct by shape; all fp32)')
    args = ap.parse_args()
    if args.head_after is None:
        args.head_after = 4 if args.dtype == 'bf16' else 8
    return args


def host_threads():
    """CPU threads this process may really use: min(affinity, cgroup quota), so that a box whose cgroup grants
    16 of 256 hardware threads is not oversubscribed 16x."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    env = os.environ.get('LCTVQA_CPU_THREADS')
    if env:
        n = int(env)
    return min(n, 64) if n > 64 else n


def conv_source_stamp():
    """Hash of the trunk-convolution kernel sources: a committed PMC profile is only quoted for the kernels it measured."""
    import hashlib
    h = hashlib.sha256()
    for f in ('conv3x3.hip', 'conv3x3_bf16.hip', 'conv3x3_wino.hip', 'conv3x3_wino4.hip', 'conv3x3_wino_wgrad.hip', 'gemm.hip', 'common.h'):
        with open(os.path.join(ROOT, 'lct_vqa_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    import inspect
    from lct_vqa_amd import ops as _ops
    h.update(inspect.getsource(_ops._conv_block).encode())      # which layers take the split Winograd path is decided there
    return h.hexdigest()[:16]


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown CPU'


def cpu_baseline(steps):
    """The oracle's W train step (fwd + CE + bwd + clip + Adam) at full resolution, B=4 and B=32, timed on
    the host cores of this box: median of `steps` steps after one warm-up each.  Reported baseline only (kind 'port': the
    reference's Python cannot travel); `value` is the better of the two batch sizes."""
    from oracle import vqa_oracle as O
    from lct_vqa_amd.basic_vqa import models
    ncores = host_threads()
    torch.set_num_threads(ncores)
    torch.manual_seed(10)
    m = models.VqaModel(**DIMS)
    P0 = {k: v.detach() for k, v in m.state_dict().items()}
    del m
    res = {}
    for B, n in ((4, steps), (32, max(3, steps // 2))):
        P = dict(P0)
        img = torch.randn(B, 3, 224, 224)
        qst = torch.randint(DIMS['qst_vocab_size'], (B, 14)); qst[:, 0] = 2
        lab = torch.randint(DIMS['ans_vocab_size'], (B,))
        state, times = {}, []
        for it in range(n + 1):
            masks = O.draw_masks(O.w_mask_shapes(B, DIMS['ans_vocab_size'], DIMS['embed_size']))
            t0 = time.perf_counter()
            _, _, P, _ = O.w_train_step(img, qst, lab, P, state, masks)
            times.append(time.perf_counter() - t0)
        times = sorted(times[1:])
        res[B] = (times[len(times) // 2], n)
    best = max(res, key=lambda B: B / res[B][0])
    return {'value': best / res[best][0], 'unit': 'pairs/s', 'cores': ncores, 'kind': 'port', 'cpu': cpu_model_name(),
            'sample': 'oracle W train step, 224x224, T=14, A=3129, QV=17858, fp32, median after 1 warm-up: '
                      + '; '.join(f'B={B}: {res[B][0] * 1e3:.0f} ms/step = {B / res[B][0]:.1f} pairs/s ({res[B][1]} steps)' for B in sorted(res))
                      + f' (torch {torch.__version__} CPU, {ncores} threads)'}


def self_launch(args):
    """N > 1 without a launcher: run ``torch.distributed.run`` as a CHILD of this process (which has not initialised HIP:
    importing torch does not) with the same arguments; the ranks inherit stdout, so rank 0's JSON line is this command's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def dp_record(world, local, reducers):
    """What lets a reader verify the N ranks of this line: transport, one entry per rank with the device it drives (gathered
    over the process group itself), bucket layout, how many bucket all-reduces the backward hooks launched per step and the
    time finish() waited for them (HIP events, timed region only).  Collective on every rank; None at world size 1."""
    if world == 1:
        return None
    import torch.distributed as dist
    from lct_vqa_amd import parallel
    # the record is diagnostic: nothing in it may cost the bench line.  The gather itself is a collective every rank takes part in, so
    # it either works on all ranks or raises on all of them (and then the record says why instead of listing devices).
    rec = {'backend': dist.get_backend(), 'world': world, 'rehearsal_one_gpu': os.environ.get('LCTVQA_REHEARSE_ONE_GPU') == '1'}
    try:
        rec['rccl'] = '.'.join(map(str, torch.cuda.nccl.version())) if rec['backend'] == 'nccl' else None
    except Exception as e:                          # noqa: BLE001
        rec['rccl'] = f'unavailable ({type(e).__name__})'
    try:
        ranks = parallel.describe_ranks(local)
        rec['devices'] = ranks
        rec['distinct_devices'] = len({(r['bus'], r['uuid'], r['device']) for r in ranks})
    except Exception as e:                          # noqa: BLE001
        rec['devices'] = f'gather failed: {type(e).__name__}: {e}'
    try:
        per = [r.describe() for r in reducers]
        for i, d in enumerate(per):
            rec.update({(k if i == 0 else f'{k}_{i}'): v for k, v in d.items()})
        rec['allreduce_ms_per_step'] = sum(d['allreduce_wait_ms_per_step'] for d in per)
    except Exception as e:                          # noqa: BLE001
        rec['reducers'] = f'unavailable: {type(e).__name__}: {e}'
    return rec


_FAMILIES = ('gemm1x1_bf16', 'wgrad_bf16', 'bn_apply_bf16', 'bn_bwd_sums_bf16', 'bn_bwd_apply_bf16', 'conv3x3_bf16', 'conv3x3_gemm_bf16',
             'wino4', 'wino', 'bf16', 'gemm1x1', 'wgrad', 'dgrad', 'first', 'gemm')


def family_rooflines(conv, steps, bf16, step_ms):
    """Group the HIP-event records of ops._timed by kernel family and price each family against ITS OWN bound: MFMA when its
    algorithmic intensity (flops / algorithmic bytes) lies above the ridge of the dtype's dense peak over the 8 TB/s HBM spec, else
    HBM.  -> (dominant family's roofline dict, [all families, by time])."""
    peak_tf = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    fam = {}
    for r in conv:
        key = next((f for f in _FAMILIES if r['name'].startswith(f)), r['name'].split('_')[0])
        a = fam.setdefault(key, {'family': key, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'launches': 0})
        # the Winograd kernels are recorded with the direct-convolution count; the MFMA pipe executes 1/4 (F(4x4,3x3)) or 1/2.25 of it
        executed = r['flops'] / (4.0 if key == 'wino4' else 2.25 if key == 'wino' else 1.0)
        a['ms'] += r['ms']; a['flops'] += executed; a['bytes'] += r.get('bytes', 0.0) or 0.0; a['launches'] += 1
    out = []
    for a in fam.values():
        if a['ms'] <= 0:
            continue
        ai = a['flops'] / a['bytes'] if a['bytes'] else float('inf')
        bound = 'mfma' if ai > ridge else 'hbm'
        ach = a['flops'] / (a['ms'] * 1e-3) / 1e12 if bound == 'mfma' else a['bytes'] / (a['ms'] * 1e-3) / 1e9
        pk = peak_tf if bound == 'mfma' else HBM_PEAK_GBS
        out.append({'kernel': a['family'], 'bound': bound, 'achieved': round(ach, 2), 'peak': pk, 'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s',
                    'frac': round(ach / pk, 4), 'traffic': None, 'launches': a['launches'], 'avg_launch_ms': round(a['ms'] / a['launches'], 4),
                    'ms_per_step': round(a['ms'] / steps, 3), 'share_of_step': round(a['ms'] / steps / step_ms, 4),
                    'flops_per_launch': a['flops'] / a['launches'], 'algorithmic_bytes_per_launch': a['bytes'] / a['launches'],
                    'algorithmic_intensity': round(ai, 1) if a['bytes'] else None})
    out.sort(key=lambda d: -d['ms_per_step'])
    return (out[0] if out else None), out


def bench_darts_vqa(args, rank, world, dev, img, qst, lab):
    """BASELINE configs[4] (fp32 form, one GPU): the weight step of the darts_vqa driver with the PC-DARTS search encoder,
    K timed steps between the barriers; the architecture step (2-level, second order) is timed once beside it."""
    import types
    from lct_vqa_amd import ops, parallel
    from lct_vqa_amd.darts_vqa import config as dconfig, factory
    from lct_vqa_amd.basic_vqa import config as bconfig
    from lct_vqa_amd.optim import FusedAdam
    dconfig.DEVICE = bconfig.DEVICE = dev
    a = types.SimpleNamespace(unified=False, embed_size=512, word_embed_size=300, num_layers=1, hidden_size=512, arch_type='darts',
                              learn_rate=1e-3, step_size=10, gamma=0.1, arch_learn_rate=6e-4, arch_wt_decay=1e-3, qst_only=False)
    ds = types.SimpleNamespace(qst_vocab=types.SimpleNamespace(vocab_size=DIMS['qst_vocab_size']),
                               ans_vocab=types.SimpleNamespace(vocab_size=DIMS['ans_vocab_size']))
    torch.manual_seed(10)
    model = factory.get_vqa_model(a, ds).to(dev).train()
    opt = FusedAdam(model.parameters(), lr=a.learn_rate, flatten=True)
    ops.grad_sink(not args.no_grad_sink)            # the fused edges add their weight gradients into the arena views themselves
    arch = factory.get_architect(a, model)
    red = parallel.FlatGradReducer(opt.flat['params'], opt.flat_grad, scale_fn=parallel.hip_scale) if world > 1 else None
    B = args.batch

    def step():
        opt.zero_grad()
        loss = model._loss(img, qst, lab, False)
        loss.backward()
        if red is not None:
            red.finish()
        opt.step(max_grad_norm=5.0)
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if red is not None:
        red.describe()
        red.stats.update(steps=0, launched_by_hooks=0, launched_by_finish=0, wait_ms=0.0)
        red.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    dp = dp_record(world, dev.index, [red] if red is not None else [])
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    arch_ms = None
    if rank == 0 and world == 1:
        arch.step(img, qst, lab, img, qst, lab, a.learn_rate, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        arch.step(img, qst, lab, img, qst, lab, a.learn_rate, opt)
        torch.cuda.synchronize()
        arch_ms = (time.perf_counter() - t1) * 1e3
    if rank == 0:
        ms = dt / args.steps * 1e3
        # algorithmic HBM bytes of the encoder's 56 MixedOp edges per pair (csrc/mixed_edge.hip: 29 T forward + 60 T backward per
        # edge, T = 4 * P * Cq bytes): cells at 224^2 x 16, 112^2 x 32 (reduction), 56^2 x 64 (reduction), 56^2 x 64
        T = [224 * 224 * 4 * 4, 112 * 112 * 8 * 4, 56 * 56 * 16 * 4, 56 * 56 * 16 * 4]
        enc_bytes = 89 * 14 * sum(T)
        bf = args.dtype == 'bf16'
        out = {'metric': 'image-question pairs/sec (darts_vqa weight step: PC-DARTS search encoder + LSTM encoder-decoder) at 224x224, seq=14, 3129-way'
                         + (' [bf16 compute dtype: bf16 head GEMMs and fused token head, fp32 accumulation and master weights; the MixedOp edges stay fp32]' if bf else ''),
               'value': round(world * B * args.steps / dt, 2), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': "darts_vqa VqaModel(img_encoder_type='darts') %s weight step (fwd + CE(ans) + CE(qst) + bwd + clip + Adam), "
                                      'batch %d per GPU, 224x224x3, 14 tokens, A=3129, QV=17858: BASELINE configs[4] on one GPU' % (args.dtype, B),
                          'global_batch': world * B, 'seq_len': 14, 'parallelism': f'dp{world}',
                          'architect_step_ms': None if arch_ms is None else round(arch_ms, 1)},
               'loss': round(float(loss.item()), 5),
               'roofline': {'bound': 'hbm', 'kernel': 'edge_fwd_* / edge_bwd_* (fused MixedOp edge kernels, 56 edges x 5 forward + 10..12 backward launches)',
                            'achieved': round(enc_bytes * B / (ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
                            'frac': round(enc_bytes * B / (ms * 1e-3) / 8e12, 4), 'traffic': None,
                            'algorithmic_bytes_per_pair': enc_bytes,
                            'note': 'algorithmic bytes of the encoder edges only, over the WHOLE step time (LSTM, token head and torch glue included): a lower bound of the edge kernels\' own rate'},
               'dp': dp, 'cpu_baseline': None}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    from lct_vqa_amd import ops, parallel
    from lct_vqa_amd._lib import lib
    from lct_vqa_amd.basic_vqa import models, config
    from lct_vqa_amd.optim import FusedAdam
    lib.load()                                      # fail loudly if the HIP library is missing
    if args.conv_algo:
        ops.set_conv_algo(args.conv_algo)
    ops.set_compute_dtype(args.dtype)
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    dev = torch.device('cuda', local)
    config.DEVICE = dev
    config.ARCH_TYPE = 'fixed'

    B = args.batch
    g = torch.Generator().manual_seed(10 + rank)    # rank r uses seed 10+r for data
    img = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    qst = torch.randint(DIMS['qst_vocab_size'], (B, 14), generator=g); qst[:, 0] = 2
    qst = qst.to(dev)
    lab = torch.randint(DIMS['ans_vocab_size'], (B,), generator=g).to(dev)
    ops.manual_seed(10 + rank)

    if args.workload == 'darts_vqa':                # its own model and optimizer: nothing of the W model is built beside it
        return bench_darts_vqa(args, rank, world, dev, img, qst, lab)

    torch.manual_seed(10)                           # config.SEED: identical replicas on every rank
    if args.workload == 'resnet_gru':
        from lct_vqa_amd.basic_vqa import models_ext
        model = models_ext.VqaModelResNetGRU(pretrained=not args.train_encoder, **DIMS).to(dev).train()
        model.img_encoder.model.cache_folded_weights = not args.train_encoder    # frozen trunk: nothing ever writes its tensors
    else:
        model = models.VqaModel(**DIMS).to(dev).train()
    opt = FusedAdam(model.trainable_parameters(), lr=config.LEARNING_RATE, flatten=True)
    ops.grad_sink(not args.no_grad_sink)            # fused backward nodes accumulate into the arena's gradient views
    trainable = opt.flat['params']
    reducer = parallel.FlatGradReducer(trainable, opt.flat_grad, scale_fn=parallel.hip_scale) if world > 1 else None

    def step():
        opt.zero_grad()
        logits = model(img, qst)
        loss = ops.cross_entropy(logits, lab)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step(max_grad_norm=config.GRAD_CLIP)
        return loss

    ef_red = None
    locals_ef_red = lambda: ef_red
    if args.workload == 'lct':
        # LCT stage 1 (EF model) + stage 2 (pseudo QA from EF.generate, W trained on real + pseudo tests)
        from lct_vqa_amd.basic_vqa import models_lct
        torch.manual_seed(11)
        # --train-encoder: the reference's --no_pretrain_enc run (config.py:71-72,93; models_lct.py:62-63): the examiner's VGG19 is
        # trained, so every step adds its full backward (dgrad + weight gradients) and generate sees the updated trunk
        ef = models_lct.VqaModel(pretrained=not args.train_encoder, **DIMS).to(dev).train()
        ef.qst_encoder.max_length = 14
        if not args.train_encoder:
            ef.img_encoder.model.cache_packed_weights = True    # pretrained=True: frozen trunk (models_lct.py:62-63 under no_grad)
        ef_opt = FusedAdam(ef.trainable_parameters(), lr=config.LEARNING_RATE, flatten=True)
        ef_red = parallel.FlatGradReducer(ef_opt.flat['params'], ef_opt.flat_grad, scale_fn=parallel.hip_scale) if world > 1 else None
        w_step = step

        def step(convs=None):  # noqa: F811
            ef_opt.zero_grad()
            # the examiner's trunk is frozen (pretrained): stage 1's forward and stage 2's generate see the same images
            # through the same weights, so its convolutional part runs once per batch (--lct-no-share: twice, as the
            # reference's two calls do)
            if convs is not None:
                conv, conv_w = convs
            else:
                conv, conv_w = (ef.img_encoder.conv_features(img) if args.lct_share_trunk and not args.train_encoder else None), None
            l1 = ef._loss(img, qst, lab, conv)
            l1.backward()
            if ef_red is not None:
                ef_red.finish()
            ef_opt.step(max_grad_norm=config.GRAD_CLIP)
            with torch.no_grad():
                pq, pa = ef.generate(img, conv)
                pa = ops.softmax_rows(pa, 1.0)
            opt.zero_grad()
            l2 = model._soft_loss(img, qst, lab, pq, pa, conv_w)
            l2.backward()
            if reducer is not None:
                reducer.finish()
            opt.step(max_grad_norm=config.GRAD_CLIP)
            return l2

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def arm_reducers():
        # statistics of the timed region only
        for r in (reducer, locals_ef_red()):
            if r is not None:
                r.describe()
                r.stats.update(steps=0, launched_by_hooks=0, launched_by_finish=0, wait_ms=0.0)
                r.timing(True)

    # The W model's VGG19 is frozen by construction (models.py:35-36; no optimizer or architect of this step holds its
    # parameters): keep its kernel-layout weights between steps instead of re-packing 15 layers per step.
    model.img_encoder.model.cache_packed_weights = True
    pipelined = (args.workload in ('w', 'resnet_gru')) and not args.no_pipeline and not args.train_encoder
    if args.head_priority and not args.no_pipeline and not args.train_encoder:
        head_stream = torch.cuda.Stream(priority=-1)
        head_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(head_stream)
    if pipelined:
        # Two-stream software pipeline.  The VGG19 trunk is frozen (models.py:35-36) so trunk(batch i+1) is independent
        # of step i's parameter update: it runs on `trunk_stream` while head fwd/bwd + all-reduce + clip + Adam of batch
        # i run on the main stream.  Every timed step still performs one trunk and one head pass; the pipeline is
        # primed and drained INSIDE the timed region (K trunks + K heads between the two barriers).
        main_stream = torch.cuda.current_stream()
        nsplit = max(1, args.trunk_split)
        stem_after = args.stem_after if nsplit == 1 else 0
        ops.set_trunk_marker(args.head_after, stem_after)
        trunk_streams = [torch.cuda.Stream() for _ in range(nsplit)]
        trunk_stream = trunk_streams[0]
        stem_stream = torch.cuda.Stream()
        img_parts = [p.contiguous() for p in img.chunk(nsplit)]
        vgg = model.img_encoder.model
        if args.workload == 'resnet_gru':
            stem_after = 0

        def launch_stem(after_event):
            # conv1_1 of the NEXT batch (HBM-write bound, 3.3 GB) on a stream of its own, released when the current trunk
            # has passed conv layer --stem-after: it runs beside MFMA-bound layers instead of in front of conv1_2
            if not stem_after:
                return None
            stem_stream.wait_stream(main_stream)
            if after_event is not None:
                stem_stream.wait_event(after_event)
            with torch.cuda.stream(stem_stream):
                out = vgg.stem(img)
                ev = torch.cuda.Event()
                ev.record(stem_stream)
            out.record_stream(trunk_stream)
            return out, ev

        def launch_trunk(stem=None):
            # The frozen trunk has no cross-sample coupling: its batch is cut into `nsplit` parts that run the 15 conv layers
            # on streams of their own, so the workgroups of one part fill the tail of the other's kernels (a layer's grid
            # is not a multiple of the 256 CUs: 12.25 rounds for the 28x28 layers) and the HBM-bound conv1_1 / the
            # classifier GEMMs of one part overlap MFMA-bound layers of the other.
            feats = []
            for st, part in zip(trunk_streams, img_parts):
                st.wait_stream(main_stream)
                with torch.cuda.stream(st):
                    if stem is not None:
                        st.wait_event(stem[1])
                        feats.append(model.img_encoder.trunk(part, stem=stem[0]))
                    elif args.classifier_with_head:
                        feats.append(model.img_encoder.conv_features(part))
                    else:
                        feats.append(model.img_encoder.trunk(part))
            for st in trunk_streams[1:]:
                trunk_stream.wait_stream(st)
            with torch.cuda.stream(trunk_stream):
                f = torch.cat(feats) if nsplit > 1 else feats[0]
                ev = torch.cuda.Event()
                ev.record(trunk_stream)
            for t in feats:
                t.record_stream(trunk_stream)
            f.record_stream(main_stream)
            return f, ev

        def run_steps(k):
            nxt = launch_trunk(launch_stem(None))
            out = None
            for i in range(k):
                feat, ev = nxt
                if i + 1 < k:
                    stem = launch_stem(ops.trunk_marker_event(stem_after))         # marker of trunk(i), launched last round
                    nxt = launch_trunk(stem)
                    if args.head_after and ops.trunk_marker_event(args.head_after) is not None:
                        # start this batch's head only when the NEXT batch's trunk has passed conv layer --head-after: the
                        # head's small kernels then share the chip with layers whose grids leave idle tails (28x28: 12.25
                        # rounds of 256 CUs) instead of landing on conv1_2 / conv2_1
                        main_stream.wait_event(ops.trunk_marker_event(args.head_after))
                main_stream.wait_event(ev)
                opt.zero_grad()
                logits = model(img, qst, conv_feature=feat) if args.classifier_with_head else model(img, qst, feat)
                out = ops.cross_entropy(logits, lab)
                out.backward()
                if reducer is not None:
                    reducer.finish()
                opt.step(max_grad_norm=config.GRAD_CLIP)
            return out

        if args.warmup:
            run_steps(args.warmup)
        ops.conv_timer_start()
        barrier()
        arm_reducers()
        t0 = time.perf_counter()
        loss = run_steps(args.steps)
        for st in trunk_streams:
            st.synchronize()
        barrier()
        dt = time.perf_counter() - t0
    elif args.workload == 'lct' and args.lct_share_trunk and not args.no_pipeline and not args.train_encoder:
        # Both trunks of the LCT step are frozen (examiner: pretrained, W: by construction), so their convolutional parts for
        # batch i+1 run on a second HIP stream while stage 1 / generate / stage 2 of batch i run on the main stream -- the
        # schedule of the W workload above.  Every timed step still contains both trunk passes and all the head work.
        main_stream = torch.cuda.current_stream()
        trunk_stream = torch.cuda.Stream()

        def launch_convs():
            trunk_stream.wait_stream(main_stream)
            with torch.cuda.stream(trunk_stream):
                cs = (ef.img_encoder.conv_features(img), model.img_encoder.conv_features(img))
                ev = torch.cuda.Event()
                ev.record(trunk_stream)
            for c in cs:
                c.record_stream(main_stream)
            return cs, ev

        def run_steps(k):
            nxt = launch_convs()
            out = None
            for i in range(k):
                cs, ev = nxt
                if i + 1 < k:
                    nxt = launch_convs()
                main_stream.wait_event(ev)
                out = step(cs)
            return out

        pipelined = True
        if args.warmup:
            run_steps(args.warmup)
        ops.conv_timer_start()
        barrier()
        arm_reducers()
        t0 = time.perf_counter()
        loss = run_steps(args.steps)
        trunk_stream.synchronize()
        barrier()
        dt = time.perf_counter() - t0
    else:
        for _ in range(args.warmup):
            step()
        ops.conv_timer_start()                          # HIP events around every conv3x3 MFMA launch (same stream)
        barrier()
        arm_reducers()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        barrier()
        dt = time.perf_counter() - t0
    conv = ops.conv_timer_stop()
    dp = dp_record(world, local, [r for r in (reducer, ef_red) if r is not None])
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    # ---- roofline of the dominant kernel family (the 15 MFMA conv launches of the VGG19 trunk per step) ----
    # achieved = ALGORITHMIC FLOPs (direct-convolution count 2*B*H*W*Cout*Cin*9) / live HIP-event
    # time of those launches.  With the Winograd kernels the MFMA pipe executes 2.25x fewer FLOPs than that count, so
    # `frac` can exceed 1; `mfma_executed_*` is what the pipe really ran (its utilisation against the dense fp32 peak):
    # F(4x4,3x3) executes 1/4 of the direct count, F(2x2,3x3) 1/2.25.
    algo = ops.get_conv_algo()
    tot_flops = sum(r['flops'] for r in conv)
    tot_exec = sum(r['flops'] / (4.0 if r['name'].startswith('wino4') else 2.25 if r['name'].startswith('wino') else 1.0) for r in conv)
    ngemm = sum(1 for r in conv if r['name'].startswith('gemm') and 'bf16' not in r['name'])
    ngemm_bf = sum(1 for r in conv if r['name'].startswith('gemm') and 'bf16' in r['name'])
    tot_ms = sum(r['ms'] for r in conv)
    nlaunch = max(1, len(conv))
    direct_equiv = tot_flops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    executed = tot_exec / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    nwino = sum(1 for r in conv if r['name'].startswith('wino'))
    nw4 = sum(1 for r in conv if r['name'].startswith('wino4'))
    nbf = sum(1 for r in conv if r['name'].startswith('bf16'))
    peak = BF16_MFMA_PEAK_TFLOPS if (nbf or ngemm_bf) else FP32_MFMA_PEAK_TFLOPS
    # achieved / frac: the multiply-adds the kernels' ALGORITHM performs on the MFMA pipe (Winograd F(4x4,3x3): 36 per
    # 4x4 output tile and channel pair = 1/4 of direct convolution's 144; F(2x2,3x3): 1/2.25) over the live HIP-event
    # time of those launches, against the dense fp32 MFMA peak: a utilisation, <= 1.  What Winograd buys on top is
    # `algorithmic_speedup`; `direct_equivalent_tflops` = the direct-convolution FLOP count per second.
    roofline = {'bound': 'mfma',
                'kernel': ('gemm_bf16_kernel (1x1 convolutions of the bottlenecks on v_mfma_f32_32x32x16_bf16, %d launches) + conv3x3_bf16_kernel (%d)' % (ngemm_bf, nbf)) if ngemm_bf else
                          ('gemm_f32_kernel (1x1 convolutions of the bottlenecks as fp32 MFMA GEMMs, %d launches) + conv3x3_wino4_kernel (%d)' % (ngemm, nw4)) if ngemm else
                          ('conv3x3_bf16_kernel (direct implicit GEMM on v_mfma_f32_32x32x16_bf16, fp32 accumulate, %d launches)' % nbf) if nbf else
                          ('conv3x3_wino4_kernel (Winograd F(4x4,3x3), %d launches, %d of them as input transform + 36 batched fp32 GEMMs + output '
                           'transform) + conv3x3_wino2/3_kernel (F(2x2,3x3), %d) + conv3x3_mfma_kernel (direct, %d): fp32 MFMA'
                           % (nw4, sum(1 for r in conv if r['name'].endswith('_split')), nwino - nw4, nlaunch - nwino)) if nwino else
                          'conv3x3_mfma_kernel (direct implicit GEMM, 15 VGG19 layers/step)',
                'algorithm': 'direct' if nbf else algo, 'achieved': round(executed, 2),
                'peak': peak, 'unit': 'TFLOP/s', 'frac': round(executed / peak, 4),
                'algorithmic_speedup': round(tot_flops / tot_exec, 3) if tot_exec else None,
                'direct_equivalent_tflops': round(direct_equiv, 2),
                'traffic': None, 'launches': nlaunch, 'avg_launch_ms': round(tot_ms / nlaunch, 4),
                'flops_per_launch': tot_exec / nlaunch, 'direct_flops_per_launch': tot_flops / nlaunch,
                'algorithmic_bytes_per_launch': sum(r.get('bytes', 0) for r in conv) / nlaunch or None,
                'conv_ms_per_step': round(tot_ms / args.steps, 3)}
    # HBM bytes per launch of the same kernel family from the committed rocprofv3 --pmc passes of this command (FETCH_SIZE,
    # WRITE_SIZE: tools/pmc_profile.py); used only while the kernels are the ones that were profiled (source stamp)
    # ... and only for the workload / dtype / batch the passes were taken on (the W workload at B=256): another workload's layers have
    # other shapes, so its algorithmic bytes and these measured bytes do not belong in one ratio.
    try:
        kind = 'bf16' if nbf else 'wino' if nwino else 'gemm'
        cands = [os.environ['LCTVQA_PMC_FILE']] if os.environ.get('LCTVQA_PMC_FILE') else \
            sorted(__import__('glob').glob(os.path.join(ROOT, 'profiles', f'r*_pmc_conv_{kind}.json')), reverse=True)
        pfile = cands[0]
        pm = json.load(open(pfile))
        shown = os.path.relpath(pfile, ROOT)
        if args.workload != 'w' or B != 256:
            roofline['traffic_source'] = f'{shown} was measured on workload w at B=256: not reported for this run'
        elif pm.get('csrc_stamp') in (None, conv_source_stamp()):
            roofline['traffic'] = pm['conv3x3_family']['hbm_bytes_per_launch']
            roofline['traffic_source'] = f'{shown} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, per launch)'
            if roofline['algorithmic_bytes_per_launch']:
                roofline['traffic_ratio'] = round(roofline['traffic'] / roofline['algorithmic_bytes_per_launch'], 3)
        else:
            roofline['traffic_source'] = f'{shown} is of other kernel sources (stamp mismatch): not reported'
    except Exception:
        pass
    if conv and (args.train_encoder or args.workload == 'resnet_gru'):
        # these steps are not the W workload's 15 convolution launches: every timed kernel family is priced against its own bound and
        # the line carries the one the step spends most of its time in
        dom, fams = family_rooflines(conv, args.steps, args.dtype == 'bf16', ms)
        if dom is not None:
            roofline = dict(dom)
            roofline['families'] = fams
            roofline['timed_share_of_step'] = round(sum(f['ms_per_step'] for f in fams) / ms, 4)
            roofline['note'] = ('HIP events on the launch stream around every launch of the family inside the timed region; algorithmic bytes = '
                                'operands + results once; untimed: statistics finishes, pools, weight conversion, optimizer')
    if args.layer_report:
        per = {}
        for r in conv:
            k = r['name']
            a = per.setdefault(k, {'ms': 0.0, 'flops': 0.0, 'n': 0})
            a['ms'] += r['ms']; a['flops'] += r['flops']; a['n'] += 1
        with open(args.layer_report, 'w') as f:
            for k, a in per.items():
                f.write(f"{k}: {a['ms'] / a['n']:.3f} ms/launch, {a['flops'] / a['ms'] / 1e9:.1f} TFLOP/s ({a['n']} launches)\n")
    out = {
        'metric': ('image-question pairs/sec (train step) at 224x224, seq=14, 3129-way' if args.workload == 'w' else
                   ('image-question pairs/sec (ResNet-152+GRU train step, %s trunk) at 224x224, seq=14, 3129-way' % ('trainable' if args.train_encoder else 'frozen')) if args.workload == 'resnet_gru' else
                   'image-question pairs/sec (LCT stage1+stage2 step: EF + generate + W soft loss%s) at 224x224, seq=14, 3129-way' % (', examiner trunk TRAINED' if args.train_encoder else ''))
                  + ((' [bf16 compute dtype: bf16-native TRAINABLE trunk (bf16 activations and activation gradients, fp32 master weights / statistics / '
                      'weight gradients) and bf16 head GEMMs: NOT the headline fp32 metric]' if args.train_encoder and args.workload == 'resnet_gru' else
                      ' [bf16 compute dtype: bf16 frozen trunk and bf16 head GEMMs, fp32 accumulate, fp32 master weights: NOT the headline fp32 metric]')
                     if args.dtype == 'bf16' else ''),
        'value': round(value, 2),
        'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': ('basic_vqa VGG19+LSTM fp32 W-model train step (frozen VGG19 trunk, fwd+CE+bwd+clip+Adam), '
                                'batch %d per GPU, 224x224x3 images, 14-token questions, 3129 answers, QV=17858' % B)
                               if args.workload == 'w' else
                               (('ResNet-152 (TRAINABLE: batch-statistics BatchNorm, full backward)' if args.train_encoder else
                                 'ResNet-152 (frozen, BatchNorm on running statistics folded into the convolutions)')
                                + ' + GRU + fusion head, %s train step (fwd+CE+bwd+clip+Adam), batch %d per GPU, 224x224x3, 14 tokens, 3129 answers, '
                                  "QV=17858: BASELINE configs[2]'s model shape on one GPU; the reference has no ResNet / GRU" % (args.dtype, B))
                               if args.workload == 'resnet_gru' else
                               ('basic_vqa LCT stage-1 + stage-2 step fp32 (experiment.py:216-268): examiner (models_lct, %s'
                                ') fwd+loss+bwd+clip+Adam, examiner.generate (14 tokens), W soft-loss '
                                'fwd+bwd+clip+Adam; batch %d per GPU, 224x224x3, QV=17858, A=3129; frozen-trunk conv features '
                                'evaluated %s' % ('VGG19 trained from scratch: --no_pretrain_enc' if args.train_encoder else 'frozen pretrained VGG19', B, 'once per model and batch (examiner: shared by stage 1 and generate; W: '
                                                     'shared by its two forwards)' if args.lct_share_trunk else
                                                  'per call for the examiner, once for W')),
                   'global_batch': world * B, 'seq_len': 14, 'parallelism': f'dp{world}',
                   'schedule': ('trunk(i+1) || head(i) on two HIP streams' + (', head starts after conv layer %d of trunk(i+1)' % args.head_after if args.head_after and args.workload == 'w' else '')
                                + (', head on a high-priority stream' if args.head_priority else '')) if pipelined else 'single stream'},
        'loss': round(float(loss.item()), 5),
        'model_flops_per_step': 39.48e9 * B if args.workload == 'w' else None,
        'model_tflops': round(39.48e9 * B / (ms * 1e-3) / 1e12, 2) if args.workload == 'w' else None,
        'roofline': roofline,
        'dp': dp,
    }
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(args.cpu_steps)
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
