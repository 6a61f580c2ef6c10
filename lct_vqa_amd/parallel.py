"""Data-parallel train step: one process per GPU, torch.distributed (backend 'nccl' == RCCL on
ROCm, xGMI point-to-point links), gradients all-reduced as a few large flat buckets that are
launched as soon as backward has produced them.

The reference has no distributed code at all; the collective sits where
``loss.backward()`` returns in basic_vqa/experiment.py:225 / :255, before ``clip_grad_norm_`` and
``optimizer.step()``.  Samples are independent in every layer of the VGG/LSTM/fusion models, so
DP-N on batch B/N each equals the single-process batch-B step up to summation order.

Works on any flat fp32 gradient buffer (CPU tensors + gloo in the unit tests, HBM + RCCL on MI355X).
Bucket size: xGMI is 7 point-to-point links x ~153 GB/s per GPU, so a ring step is per-link bound
and wants few, large messages: default 32 MiB buckets (the W model's 84 MB of gradients = 3 buckets).
"""
import torch
import torch.distributed as dist


class FlatGradReducer:
    def __init__(self, params, flat_grad, process_group=None, bucket_bytes=32 << 20, scale_fn=None):
        """params: parameters whose .grad are views into ``flat_grad`` in ascending order (optim.FusedAdam(flatten=True)'s
        arena, which pads every parameter to a 16-byte boundary, or ``make_flat_grad``).  A bucket is a contiguous range of
        the buffer (padding included: it is zero on every rank)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat = flat_grad
        self.scale_fn = scale_fn
        self.buckets = []          # [start, end, n_params_pending, n_params]
        self.param_bucket = {}
        per = max(1, bucket_bytes // flat_grad.element_size())
        base, size = flat_grad.data_ptr(), flat_grad.element_size()
        start, ids, prev_end = 0, [], 0
        for p in params:
            if p.grad is None:
                raise ValueError('FlatGradReducer: a parameter has no .grad view into the flat buffer')
            o = (p.grad.data_ptr() - base) // size
            if o < prev_end or o + p.numel() > flat_grad.numel():
                raise ValueError('FlatGradReducer: parameter gradients must be ascending, disjoint views of flat_grad')
            prev_end = o + p.numel()
            ids.append(p)
            if prev_end - start >= per:
                self._close(start, prev_end, ids)
                start, ids = prev_end, []
        if ids or start < flat_grad.numel():
            if ids:
                self._close(start, flat_grad.numel(), ids)
            else:                                                  # trailing padding joins the last bucket
                self.buckets[-1][1] = flat_grad.numel()
        self.handles = []
        self._hooks = []
        # what a reader of the bench line needs to see that the collective ran: launches issued by the hooks (during
        # backward) vs by finish() (stragglers), and the time finish() spent waiting (``timing(True)``: HIP events on the
        # current stream around the waits when the buffer is in HBM, host clock otherwise)
        self.stats = {'steps': 0, 'launched_by_hooks': 0, 'launched_by_finish': 0, 'wait_ms': 0.0}
        self._timing, self._events = False, []
        if self.world > 1:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def timing(self, on=True):
        self._timing = bool(on)
        return self

    def describe(self):
        """Summary for a benchmark record.  Drains the recorded events (synchronises)."""
        if self._events:
            torch.cuda.synchronize()
            self.stats['wait_ms'] += sum(a.elapsed_time(b) for a, b in self._events)
            self._events = []
        st = self.stats
        n = max(1, st['steps'])
        return {'buckets': len(self.buckets), 'bucket_bytes': [(b[1] - b[0]) * self.flat.element_size() for b in self.buckets],
                'steps': st['steps'], 'launched_during_backward_per_step': st['launched_by_hooks'] / n,
                'launched_by_finish_per_step': st['launched_by_finish'] / n, 'allreduce_wait_ms_per_step': st['wait_ms'] / n}

    def _close(self, start, end, ids):
        b = len(self.buckets)
        self.buckets.append([start, end, len(ids), len(ids)])
        for p in ids:
            self.param_bucket[p] = b

    def reset(self):
        for b in self.buckets:
            b[2] = b[3]
        self.handles = []

    def _on_grad(self, p):
        b = self.buckets[self.param_bucket[p]]
        b[2] -= 1
        if b[2] < 0:
            raise RuntimeError('FlatGradReducer: a gradient hook fired twice for one parameter before finish() (two backward() calls, or a '
                               "parameter used by two graphs): the bucket's all-reduce is already in flight on the first pass's values.  "
                               'Call finish() after every backward(), or accumulate locally and reduce once.')
        if b[2] == 0:
            self.stats['launched_by_hooks'] += 1
            self._launch(b)

    def _launch(self, b):
        h = dist.all_reduce(self.flat[b[0]:b[1]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append(h)

    def finish(self):
        """Wait for every bucket (launching the ones whose hooks did not all fire, e.g. unused
        parameters), then average."""
        if self.world > 1:
            ev = None
            if self._timing:
                if self.flat.is_cuda:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                else:
                    import time
                    t0 = time.perf_counter()
            for b in self.buckets:
                if b[2] > 0:
                    self.stats['launched_by_finish'] += 1
                    self._launch(b)
            for h in self.handles:
                h.wait()
            if self._timing:
                if ev is not None:
                    ev[1].record()
                    self._events.append(ev)
                else:
                    self.stats['wait_ms'] += (time.perf_counter() - t0) * 1e3
            self.stats['steps'] += 1
            if self.scale_fn is not None:
                self.scale_fn(1.0 / self.world, self.flat)
            else:
                self.flat.mul_(1.0 / self.world)
        self.reset()


def hip_scale(a, flat):
    from ._lib import lib
    lib('scale', float(a), flat, flat.numel())


def init_distributed():
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract), bind this process to its GPU and bring up RCCL
    (backend 'nccl').  Returns (rank, world, local_rank).  There is ONE way to a different transport: the explicit one-GPU
    rehearsal ``LCTVQA_REHEARSE_ONE_GPU=1`` (every rank on device 0, gloo: RCCL refuses two ranks on one device; it exercises
    hooks / buckets / barriers, not xGMI).  If RCCL cannot initialise, or its first all-reduce does not return the right sum,
    this raises with RCCL's own error -- never a quiet fallback."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = 'nccl'
    if os.environ.get('LCTVQA_REHEARSE_ONE_GPU') == '1':
        local, backend = 0, 'gloo'
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl' and local >= torch.cuda.device_count():
            raise RuntimeError(f'rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPUs are visible '
                               '(one process per GPU; set LCTVQA_REHEARSE_ONE_GPU=1 only to rehearse the control flow on one device)')
        torch.cuda.set_device(local)
        try:
            if backend == 'nccl':
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            probe = torch.full((4,), float(rank + 1), device=torch.device('cuda', local))
            dist.all_reduce(probe, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            want = world * (world + 1) / 2.0
            if not bool((probe == want).all()):
                raise RuntimeError(f'first all-reduce returned {probe.tolist()}, expected {want}')
        except Exception as e:
            raise RuntimeError(f'rank {rank}/{world}: the {"RCCL (nccl)" if backend == "nccl" else backend} process group did not '
                               f'come up on cuda:{local}: {type(e).__name__}: {e}') from e
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def describe_ranks(local):
    """[{rank, device, name, bus}] gathered from every rank (rank order): which GPU each process really drives."""
    import os
    props = torch.cuda.get_device_properties(local)
    try:
        uuid = str(props.uuid)
    except Exception:                               # noqa: BLE001 - not every build exposes it
        uuid = None
    me = {'rank': dist.get_rank() if dist.is_initialized() else 0, 'device': f'cuda:{local}', 'name': props.name,
          'bus': getattr(props, 'pci_bus_id', None), 'uuid': uuid, 'pid': os.getpid()}
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out


def broadcast_buffers(model, src=0, process_group=None):
    """Copy rank ``src``'s buffers (BatchNorm running statistics, num_batches_tracked) to every rank.  Data-parallel training of
    the BatchNorm encoders (PC-DARTS search network, trainable ResNet) keeps per-rank statistics (bilevel.data_parallel's note):
    call this before validation or checkpointing so that every rank evaluates / saves the same model.  No-op at world size 1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for b in model.buffers():
        dist.broadcast(b, src=src, group=process_group)


_sync_bn = {'on': False, 'group': None}


def set_sync_bn(on=True, process_group=None):
    """SyncBN for data-parallel training of the BatchNorm trunks (BN caveat; the reference's BatchNorms are
    pcdarts/operations.py:29,43,58,62,98 -- one process, so "the batch" is everything it is given).  ON: the trainable ResNet trunk
    (resnet_ops.BnActFn in fp32, resnet_bf16 in bf16) all-reduces the per-channel fp64 sums -- (sum x, sum x^2) in the forward,
    (sum g, sum g xhat) in the backward -- between its statistics kernels and their finish, so DP-N on batches of B/N equals ONE
    process on the batch B (running statistics included, identical on every rank); dgamma / dbeta stay per-rank sums, the
    gradient reducer averages them like every other parameter gradient.  OFF (default): per-rank statistics = N reference
    processes with averaged gradients.  The fused PC-DARTS edges take their statistics inside the edge kernels and are not
    covered (per-rank statistics; parallel.broadcast_buffers aligns the running estimates).  Returns the previous setting."""
    prev = _sync_bn['on']
    _sync_bn['on'], _sync_bn['group'] = bool(on), process_group
    return prev


def sync_bn_world():
    """Ranks whose BatchNorm statistics are pooled (1 = SyncBN off or no process group)."""
    if _sync_bn['on'] and dist.is_available() and dist.is_initialized():
        return dist.get_world_size(_sync_bn['group'])
    return 1


def sync_bn_all_reduce(t):
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_sync_bn['group'])
    return t


def make_flat_grad(params):
    """Allocate one flat fp32 gradient buffer and point every p.grad at its slice (any device).
    optim.FusedAdam(flatten=True) does the same for the HBM arena; this helper serves callers that
    bring their own optimizer (and the CPU/gloo tests)."""
    params = list(params)
    n = sum(p.numel() for p in params)
    flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
    o = 0
    for p in params:
        k = p.numel()
        p.grad = flat[o:o + k].view_as(p.data)
        o += k
    return flat
