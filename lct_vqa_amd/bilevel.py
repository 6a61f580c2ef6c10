"""Shared machinery of the architecture-gradient steps (the callers of the hot path that SURVEY.md section 8 lists as
a-15: reference basic_vqa/pcdarts/architect.py:12-109, architect_lct.py:13-235, darts_vqa/pcdarts/architect_vqa.py:10-145).

What those steps need, expressed once:

* ``grads``          d loss / d tensors with unreached tensors as zeros, and the reference's check of HOW MANY were unreached
                     (36 for the W model with its frozen trunk, 2 for EF' under W's soft loss, 6 for ``qst_only``, else 0);
* ``virtual_step``   a twin of a model one plain SGD step ahead, theta' = theta - lr * g (the reference's momentum and
                     weight-decay terms are identically zero: its ``network_momentum`` / ``*_weight_decay`` are 0);
* ``fd_grad``        a central finite difference of a gradient-valued closure along a direction in parameter space,
                     (f(theta + R v) - f(theta - R v)) / 2R with R = r / ||v||, the Hessian-vector products of DARTS.

The reference does this on concatenated copies of every parameter (``_concat``: two 622 MB vectors per unrolled W
model).  Here nothing is concatenated: the twin comes from the model's own ``new()`` + ``load_state_dict`` and the
parameter-space updates are one multi-tensor launch each, in place on the live parameters (restored afterwards) --
exactly the in-place ``p.data`` writes the kernels are required to observe, which is why no kernel
caches anything derived from a trainable parameter.
"""
import contextlib
import functools

import torch
import torch.distributed as dist


@contextlib.contextmanager
def exact_edge_storage():
    """The fused MixedOp edges keep their pre-BatchNorm tensors in fp32 inside this block, whatever the compute dtype
    (darts_ops.set_edge_pre_bf16).  The architecture steps difference two gradients taken 2R apart in parameter space (R = 0.01 / ||v||,
    architect.py:89-109): a tensor stored with 8 bits of mantissa turns that difference into rounding noise."""
    from . import darts_ops
    prev = darts_ops.set_edge_pre_bf16(False)
    try:
        yield
    finally:
        darts_ops.set_edge_pre_bf16(prev)


def architecture_pass(fn):
    """Decorator of the architects' ``compute_grads``: everything they evaluate runs under ``exact_edge_storage``."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        with exact_edge_storage():
            return fn(*args, **kwargs)
    return wrapped


_dp = {'group': None, 'on': None}      # on: None = automatic (on whenever a process group with more than one rank exists)


def data_parallel(on=True, process_group=None):
    """Under data parallelism every gradient the architecture steps take (dtheta of the virtual steps, grad_wprime, kappa,
    both halves of every finite difference: SURVEY.md section 8e / 2.4) must be the gradient of the GLOBAL batch.  With this
    on, ``grads`` averages its result over the ranks (one flat all-reduce per call), so every quantity derived from it -- the
    virtual steps, ||v|| in R = r / ||v||, the alpha gradient -- is identical on all ranks.  It is ON BY DEFAULT whenever
    ``torch.distributed`` is initialised with more than one rank (an architect stepping on local-shard gradients would let
    alpha drift apart between the ranks); ``data_parallel(False)`` switches it off explicitly, ``data_parallel(None)`` restores
    the automatic setting.  Returns the previous setting.

    What it equals: for models WITHOUT BatchNorm (VGG / LSTM / fusion head) the single-process step on the concatenated batch
    (losses are means over samples; equal shards).  For the PC-DARTS and trainable ResNet encoders it equals N reference
    processes whose gradients are averaged: BatchNorm uses each rank's own batch statistics (the reference's semantics per
    process; ``parallel.set_sync_bn(True)`` pools the per-channel sums of all ranks instead -- the PC-DARTS network then runs its
    per-primitive kernels), and
    the running statistics are per rank -- ``parallel.broadcast_buffers(model)`` copies rank 0's to every rank (before a
    validation pass or a checkpoint)."""
    prev = _dp['on']
    _dp['on'], _dp['group'] = (None if on is None else bool(on)), process_group
    return prev


def _average_over_ranks(tensors):
    world = dist.get_world_size(_dp['group']) if (_dp['on'] is not False and dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return tensors
    # Two guard words ride along in the same all-reduce: the call counter and the element count.  Ranks that make different
    # numbers of grads() calls, or calls of different sizes, see sums that are not world x their own value and raise instead of
    # pairing unrelated collectives (a rank that makes NO matching call still blocks in the collective: every rank must run the
    # architect step -- INTEGRATION.md, "Data parallel").
    _dp['calls'] = _dp.get('calls', 0) + 1
    n = sum(t.numel() for t in tensors)
    ref = tensors[0]
    guard = torch.tensor([float(_dp['calls'] % 4096), float(n % 16777216)], dtype=ref.dtype, device=ref.device)
    flat = torch.cat([guard] + [t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_dp['group'])
    got = flat[:2].tolist()
    if got[0] != world * float(_dp['calls'] % 4096) or got[1] != world * float(n % 16777216):
        raise RuntimeError(f'bilevel.grads under data parallelism: the ranks are not making the same sequence of calls (call #{_dp["calls"]}, '
                           f'{n} elements here; the group summed to {got}).  Every rank must run the same architect step, or switch the '
                           'averaging off with bilevel.data_parallel(False).')
    flat = flat[2:]
    flat.mul_(1.0 / world)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view_as(t))
        o += t.numel()
    return out


def grads(loss, tensors, expect_unused=None, weight_grads=True):
    """Gradients of ``loss`` w.r.t. ``tensors`` as a list; tensors the graph does not reach get zeros.  When
    ``expect_unused`` is given, the number of unreached tensors must equal it (the reference asserts these counts).
    Averaged over the ranks when ``data_parallel`` is on.  ``weight_grads=False``: ``tensors`` holds architecture parameters only, the
    PC-DARTS ops skip their convolution-weight gradients in this backward pass (darts_ops.weight_grads)."""
    tensors = list(tensors)
    from . import ops
    with ops.no_grad_sink():                                     # autograd.grad wants the gradients as return values
        if weight_grads:
            got = torch.autograd.grad(loss, tensors, allow_unused=True)
        else:
            from . import darts_ops
            with darts_ops.weight_grads(False):
                got = torch.autograd.grad(loss, tensors, allow_unused=True)
    unused = sum(g is None for g in got)
    if expect_unused is not None and unused != expect_unused:
        raise AssertionError(f'{unused} tensors received no gradient, expected {expect_unused}')
    return _average_over_ranks([torch.zeros_like(t) if g is None else g for g, t in zip(got, tensors)])


def global_norm(vectors):
    """||concat(vectors)||_2 without concatenating."""
    return torch.linalg.vector_norm(torch.stack(torch._foreach_norm(list(vectors))))


@torch.no_grad()
def shift_(params, direction, alpha):
    """params += alpha * direction, in place and behind autograd's back (one multi-tensor launch)."""
    torch._foreach_add_([p.data for p in params], list(direction), alpha=alpha)


_CHUNK = 64 * 1024


def copy_tensors(dst, src, cache=None, key=None):
    """dst[i] <- src[i] for two lists of same-shaped, same-dtype tensors.  Contiguous tensors on a ROCm device go through ONE
    launch of ``lctvqa_multi_copy`` (torch's multi-tensor copy falls back to one copy kernel per tensor here: 2151 launches for a
    PC-DARTS examiner's state_dict); the chunk table lives in HBM and is kept in ``cache[key]`` while every address stays the
    same.  Anything else (CPU tensors, strided views) takes ``torch._foreach_copy_``."""
    dst, src = list(dst), list(src)
    if not dst:
        return
    fast = all(a.is_cuda and b.is_cuda and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape == b.shape
               for a, b in zip(dst, src))
    if not fast:
        for flt in (True, False):                                # one multi-tensor copy per dtype family
            d = [t for t in dst if t.is_floating_point() == flt]
            s = [t for t in src if t.is_floating_point() == flt]
            if d:
                torch._foreach_copy_(d, s)
        return
    from ._lib import lib
    sig = tuple((a.data_ptr(), b.data_ptr(), a.numel() * a.element_size()) for a, b in zip(dst, src))
    ent = cache.get(key) if (cache is not None and key is not None) else None
    if ent is None or ent[0] != sig:
        rows = []
        for (dp, sp, nbytes) in sig:
            for off in range(0, nbytes, _CHUNK):
                rows.append((sp + off, dp + off, min(_CHUNK, nbytes - off)))
        table = torch.tensor(rows, dtype=torch.int64).to(dst[0].device)
        ent = (sig, table, len(rows))
        if cache is not None and key is not None:
            cache[key] = ent
    if ent[2]:
        lib('multi_copy', ent[1], ent[2])


def virtual_step(model, g, lr, cache=None, key='twin'):
    """An independent copy of ``model`` (its own ``new()``: same class, same device, architecture parameters included)
    whose parameters are theta - lr * g; buffers are copied as they are.

    ``cache`` (a dict owned by the caller, e.g. an architect): the twin of the previous step is re-used instead of being
    constructed again -- the reference builds a fresh model per step, which for the W model means initialising and uploading
    a 575 MB VGG19 every time.  A re-used twin is refilled completely (every state_dict tensor and the architecture
    parameters, a handful of multi-tensor copies), and nothing else of it is ever changed by the architects, so it is
    indistinguishable from a fresh ``new()`` + ``load_state_dict``."""
    twin = cache.get(key) if cache is not None else None
    src = model.state_dict()
    if twin is not None:
        dst = twin.state_dict()
        if list(dst) != list(src) or any(a.shape != b.shape for a, b in zip(dst.values(), src.values())):
            twin = None                                          # another model: start over
    if twin is None:
        twin = model.new()
        twin.load_state_dict(src)
        if cache is not None:
            cache[key] = twin
    else:
        with torch.no_grad():
            d, s_ = list(dst.values()), list(src.values())
            if hasattr(model, 'arch_parameters'):
                try:
                    d += [a.data for a in twin.arch_parameters()]
                    s_ += [a.data for a in model.arch_parameters()]
                except AttributeError:                           # a model without a search encoder
                    pass
            copy_tensors(d, s_, cache, ('copy-table', key))
    shift_(list(twin.parameters()), g, -lr)
    return twin


def fd_grad(params, direction, closure, r=1e-2):
    """Central difference of the list-valued ``closure()`` along ``direction`` in the space of ``params``.  The closure
    is evaluated with the live parameters moved to theta + R v and theta - R v (R = r / ||v||); they are back at theta
    (up to fp32 rounding of the three in-place adds, like the reference) when this returns."""
    params, direction = list(params), list(direction)
    R = float(r / global_norm(direction))
    shift_(params, direction, R)
    plus = closure()
    shift_(params, direction, -2.0 * R)
    minus = closure()
    shift_(params, direction, R)
    return [(a - b) / (2.0 * R) for a, b in zip(plus, minus)]


def set_grads(tensors, values):
    """tensor.grad = value (copied), whether or not a .grad exists already."""
    for t, v in zip(tensors, values):
        if t.grad is None:
            t.grad = v.detach().clone()
        else:
            t.grad.detach().copy_(v)


def arch_optimizer(arch_parameters, lr, weight_decay):
    """Adam(betas=(0.5, 0.999)) on the architecture parameters (252 floats: host-side torch.optim is the right tool)."""
    return torch.optim.Adam(arch_parameters, lr=lr, betas=(0.5, 0.999), weight_decay=weight_decay)
