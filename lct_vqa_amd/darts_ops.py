"""autograd.Function wrappers of the PC-DARTS primitives (csrc/darts.hip) on NHWC fp32 activations
[B,H,W,C], plus ``run_seq`` which executes a reference-shaped ``nn.Sequential`` of standard torch
modules (ReLU / Conv2d / BatchNorm2d / AvgPool2d / MaxPool2d used purely as parameter+buffer holders,
so state_dict keys equal the reference's) through those Functions."""
import contextlib
import ctypes
import os

import torch
import torch.nn as nn

from . import ops
from ._lib import lib, _current_raw_stream
from .ops import _chk, _workspace, colsum  # noqa: F401


_scratch_pool = {}


def _scratch(n, dtype, dev):
    """Scratch that lives only inside ONE C call (per-block partial sums a finish kernel of the same call consumes): one
    growing buffer per (device, dtype, stream) instead of an allocation per launch.  Launches on a stream run in order, so the
    next user overwrites it only after the previous one has finished with it."""
    key = (dev.index, dtype, _current_raw_stream())
    buf = _scratch_pool.get(key)
    if buf is None or buf.numel() < n:
        buf = _scratch_pool[key] = torch.empty(max(int(n), 1 << 16), dtype=dtype, device=dev)
    return buf


def _gemm_ex(A, a_ms, a_ks, Bm, b_ns, b_ks, M, N, K, relu_ab):
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    need = lib.query('gemm_workspace_bytes', M, N, K)
    ws = _workspace(need, A.device) if need else None
    lib('gemm_f32_ex', A, a_ms, a_ks, Bm, b_ns, b_ks, C, N, M, N, K, None, None, 0, 0, None, 0, 1.0, ws, need, relu_ab)
    return C


class SliceChannelsFn(torch.autograd.Function):
    """x[..., c0:c0+cn] as a dense tensor (model_search.py:44-45)."""

    @staticmethod
    def forward(ctx, x, c0, cn):
        x = _chk(x)
        B, H, W, C = x.shape
        y = torch.empty((B, H, W, cn), dtype=torch.float32, device=x.device)
        lib('slice_channels', x, C, c0, y, cn, B * H * W)
        ctx.c0, ctx.C = c0, C
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _chk(dy)
        B, H, W, cn = dy.shape
        dx = torch.empty((B, H, W, ctx.C), dtype=torch.float32, device=dy.device)
        lib('slice_channels_bwd', dy, cn, ctx.c0, dx, ctx.C, B * H * W)
        return dx, None, None


class DwConvFn(torch.autograd.Function):
    """(ReLU ->) depthwise conv (operations.py:41,56,60)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, dil, relu_in):
        x, w = _chk(x), _chk(w)
        B, H, W, C = x.shape
        K = w.shape[-1]
        Ho = (H + 2 * pad - dil * (K - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (K - 1) - 1) // stride + 1
        y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
        lib('dwconv_fwd', x, w, y, B, H, W, C, K, stride, pad, dil, int(relu_in))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, int(relu_in))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil, relu_in = ctx.cfg
        dy = _chk(dy)
        B, H, W, C = x.shape
        K = w.shape[-1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            lib('dwconv_bwd_input', dy, w, x, dx, B, H, W, C, K, stride, pad, dil, relu_in)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            n = lib.query('dwconv_bwd_weight_scratch_elems', B, H, W, C, K, stride, pad, dil)
            scratch = _scratch(n, torch.float32, x.device)
            lib('dwconv_bwd_weight', dy, x, dw, scratch, B, H, W, C, K, stride, pad, dil, relu_in)
        return dx, dw, None, None, None, None


# Cout * Cin from which a 1x1 convolution (forward and input gradient) goes to the MFMA GEMM instead of the thread-per-pixel kernel: the
# preprocess layers' 64 -> 32 / 128 -> 64 products are LDS-bound there (every weight a broadcast read); encoder forward+backward at B=64
# 73.0 -> 71.2 ms, at B=16 25.2 -> 24.9 ms (tools/bench_darts.py with LCTVQA_PW_GEMM_FROM = 2^30 / 1024)
_PW_GEMM_FROM = int(os.environ.get('LCTVQA_PW_GEMM_FROM', 1024))


class PwConvFn(torch.autograd.Function):
    """(ReLU ->) 1x1 conv, no bias (operations.py:28,42,57,61) as a strided GEMM over pixels."""

    @staticmethod
    def forward(ctx, x, w, relu_in):
        x, w = _chk(x), _chk(w)
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        P = B * H * W
        if lib.query('pwconv_small_supported', P, Cout, Cin) and Cout * Cin < _PW_GEMM_FROM:
            y = torch.empty((P, Cout), dtype=torch.float32, device=x.device)
            lib('pwconv_small', x, w, None, y, P, Cout, Cin, int(bool(relu_in)), 0)
        else:
            y = _gemm_ex(x, Cin, 1, w, Cin, 1, P, Cout, Cin, 1 if relu_in else 0)
        ctx.save_for_backward(x, w)
        ctx.relu_in = bool(relu_in)
        return y.reshape(B, H, W, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _chk(dy)
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        P = B * H * W
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if lib.query('pwconv_small_supported', P, Cout, Cin) and Cout * Cin < _PW_GEMM_FROM:
                dx = torch.empty((P, Cin), dtype=torch.float32, device=x.device)
                lib('pwconv_small', dy, w, x if ctx.relu_in else None, dx, P, Cout, Cin, 0, 1)    # dY . W, * (x > 0) fused
            else:
                dx = _gemm_ex(dy, Cout, 1, w, 1, Cin, P, Cin, Cout, 0)        # dY . W
                if ctx.relu_in:
                    out = torch.empty_like(dx)
                    lib('act_bwd', dx, x, None, 1.0, out, dx.numel(), 1)      # * (x > 0)
                    dx = out
            dx = dx.reshape(B, H, W, Cin)
        if ctx.needs_input_grad[1] and _WEIGHT_GRADS:
            if lib.query('pwconv_wgrad_small_supported', P, Cout, Cin):
                dw = torch.empty_like(w)
                scratch = _scratch(lib.query('pwconv_wgrad_small_scratch_elems', P, Cout, Cin), torch.float32, x.device)
                lib('pwconv_wgrad_small', dy, x, dw, scratch, P, Cout, Cin, int(ctx.relu_in))
            else:
                dw = _gemm_ex(dy, 1, Cout, x, 1, Cin, Cout, Cin, P, 2 if ctx.relu_in else 0).reshape(w.shape)
        return dx, dw, None


def _bn_forward(x, gamma, beta, run_mean, run_var, tracked, training, momentum, eps):
    """One lctvqa_batchnorm_fwd call on an NHWC tensor: (y, mean, invstd).  ``tracked``: the holder's int64
    num_batches_tracked, incremented by the same call in training mode."""
    C = x.shape[-1]
    P = x.numel() // C
    dev = x.device
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    scratch = _scratch(lib.query('bn_scratch_doubles', P, C), torch.float64, dev)
    lib('batchnorm_fwd', x, C, y, C, P, C, gamma, beta, run_mean, run_var, tracked if training else None, mean, invstd, scratch,
        int(training), float(momentum), float(eps))
    return y, mean, invstd


def _bn_backward(dy, x, mean, invstd, gamma, training):
    C = x.shape[-1]
    P = x.numel() // C
    dev = x.device
    dx = torch.empty_like(x)
    sums = _scratch(2 * C, torch.float32, dev)
    scratch = _scratch(lib.query('bn_scratch_doubles', P, C), torch.float64, dev)
    dg = db = None
    if gamma is not None:
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
    lib('batchnorm_bwd', dy, C, x, C, mean, invstd, gamma, dx, dg, db, sums, scratch, P, C, int(training))
    return dx, dg, db


_unit = {}


def _unit_affine(C, dev):
    """(ones[C], zeros[C]) standing in for the gamma / beta of an affine=False BatchNorm where a kernel wants the pointers"""
    key = (C, dev)
    if key not in _unit:
        _unit[key] = (torch.ones(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev))
    return _unit[key]


def sync_bn_active():
    """True while parallel.set_sync_bn is on in a process group of more than one rank: the PC-DARTS layers then run their per-primitive
    kernels (no fused edge, no fused conv + BatchNorm node), whose BatchNormFn pools the per-channel sums of all ranks."""
    from . import parallel
    return parallel.sync_bn_world() > 1


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d: batch statistics + running update in train mode, running statistics in eval.  Under SyncBN (parallel.set_sync_bn,
    more than one rank, train mode) the statistics -- and the two sums of the backward -- are those of the GLOBAL batch: the kernels of
    the ResNet trunks' SyncBN (lctvqa_bn_wide_sums / _fwd_from_sums / _bwd_from_sums) with one all-reduce of [C][2] fp64 sums each way."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, tracked, training, momentum, eps):
        x = _chk(x)
        from . import parallel
        world = parallel.sync_bn_world() if training else 1
        C = x.shape[-1]
        if world > 1:
            if C % 4 != 0:
                raise NotImplementedError(f'SyncBN of a BatchNorm2d({C}): the pooled-sum kernels take channel counts that are multiples of 4')
            P, dev = x.numel() // C, x.device
            ones, zeros = _unit_affine(C, dev)
            y = torch.empty_like(x)
            stats = torch.empty((2, C), dtype=torch.float32, device=dev)
            part = _scratch(lib.query('bn_wide_blocks', P) * C * 2, torch.float64, dev)
            sums = torch.empty((C, 2), dtype=torch.float64, device=dev)
            lib('bn_wide_sums', x, None, None, None, None, P, C, 0, part, sums)
            parallel.sync_bn_all_reduce(sums)
            lib('bn_wide_fwd_from_sums', x, sums, float(P * world), gamma if gamma is not None else ones, beta if beta is not None else zeros,
                run_mean, run_var, tracked, None, 0, y, stats[0], stats[1], P, C, float(momentum), float(eps))
            mean, invstd = stats[0], stats[1]
        else:
            y, mean, invstd = _bn_forward(x, gamma, beta, run_mean, run_var, tracked, training, momentum, eps)
        ctx.save_for_backward(x, mean, invstd, gamma)
        ctx.training, ctx.world = int(training), world
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, gamma = ctx.saved_tensors
        dy = _chk(dy)
        if ctx.world > 1:
            from . import parallel
            C = x.shape[-1]
            P, dev = x.numel() // C, x.device
            ones, _ = _unit_affine(C, dev)
            dx = torch.empty_like(x)
            dg = torch.empty(C, dtype=torch.float32, device=dev)
            db = torch.empty(C, dtype=torch.float32, device=dev)
            sums = _scratch(2 * C, torch.float32, dev)
            part = _scratch(lib.query('bn_wide_blocks', P) * C * 2, torch.float64, dev)
            loc = torch.empty((C, 2), dtype=torch.float64, device=dev)
            lib('bn_wide_sums', x, dy, None, mean, invstd, P, C, 1, part, loc)
            glob = loc.clone()
            parallel.sync_bn_all_reduce(glob)
            lib('bn_wide_bwd_from_sums', dy, None, x, mean, invstd, gamma if gamma is not None else ones, loc, glob, float(P * ctx.world), dx, None,
                dg, db, sums, P, C)
            return dx, (dg if gamma is not None else None), (db if gamma is not None else None), None, None, None, None, None, None
        dx, dg, db = _bn_backward(dy, x, mean, invstd, gamma, ctx.training)
        return dx, dg, db, None, None, None, None, None, None


class Pool3x3Fn(torch.autograd.Function):
    """MaxPool2d(3,s,1) (kind 0) / AvgPool2d(3,s,1,count_include_pad=False) (kind 1), operations.py:6-7."""

    @staticmethod
    def forward(ctx, x, stride, kind):
        x = _chk(x)
        B, H, W, C = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
        lib('pool3x3_fwd', x, y, B, H, W, C, stride, kind)
        ctx.save_for_backward(x)
        ctx.cfg = (stride, kind)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        stride, kind = ctx.cfg
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        lib('pool3x3_bwd', _chk(dy), x, dx, B, H, W, C, stride, kind)
        return dx, None, None


class Subsample2Fn(torch.autograd.Function):
    """relu(x)[:, 0::2, 0::2] and relu(x)[:, 1::2, 1::2] (FactorizedReduce, operations.py:100-102)."""

    @staticmethod
    def forward(ctx, x, relu):
        x = _chk(x)
        B, H, W, C = x.shape
        y0 = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        y1 = torch.empty_like(y0)
        lib('subsample2_fwd', x, y0, B, H, W, C, 0, int(relu))
        lib('subsample2_fwd', x, y1, B, H, W, C, 1, int(relu))
        ctx.save_for_backward(x)
        ctx.relu = int(relu)
        ctx.set_materialize_grads(False)
        return y0, y1

    @staticmethod
    def backward(ctx, d0, d1):
        (x,) = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        lib('subsample2_bwd', _chk(d0) if d0 is not None else None, _chk(d1) if d1 is not None else None, x, dx,
            B, H, W, C, ctx.relu)
        return dx, None


def _ptr_array(tensors):
    return (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])


class MixedCombineFn(torch.autograd.Function):
    """state (+)= beta * channel_shuffle(cat(sum_k w_k branch_k, bypass(x)), 4)  (model_search.py:41-55,87).
    ``prev`` (or None) is updated IN PLACE and returned; ``reduce`` = stride-2 edge (bypass = MaxPool2d(2,2))."""

    @staticmethod
    def forward(ctx, prev, x, wsel, beta, reduce, *branches):
        x, wsel, beta = _chk(x), _chk(wsel), _chk(beta)
        branches = [_chk(b) for b in branches]
        B, H, W, C = x.shape
        Ho, Wo = (H // 2, W // 2) if reduce else (H, W)
        if prev is None:
            out = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
            acc = 0
        else:
            out = prev
            acc = 1
            ctx.mark_dirty(prev)
        lib('mixed_combine_fwd', _ptr_array(branches), len(branches), wsel, beta, x, B, Ho, Wo, C, int(reduce), out,
            out.stride(2), acc)
        ctx.save_for_backward(x, wsel, beta, *branches)
        ctx.reduce, ctx.has_prev = int(reduce), prev is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wsel, beta = ctx.saved_tensors[:3]
        branches = list(ctx.saved_tensors[3:])
        dout = _chk(dout)
        B, H, W, C = x.shape
        Ho, Wo = (H // 2, W // 2) if ctx.reduce else (H, W)
        P, Cq, dev = B * Ho * Wo, C // 4, x.device
        ldd = dout.stride(2)
        dx = torch.empty_like(x)
        lib('mixed_combine_bwd_x', dout, ldd, beta, x, dx, B, H, W, C, ctx.reduce)
        dbr = []
        for k, b in enumerate(branches):
            g = torch.empty_like(b)
            lib('mixed_combine_bwd_branch', dout, ldd, wsel, k, beta, g, P, Cq)
            dbr.append(g)
        dw = torch.zeros(max(len(branches), 1), dtype=torch.float32, device=dev)
        dbeta = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = _scratch(lib.query('mixed_combine_scratch_elems', B, Ho, Wo), torch.float32, dev)
        lib('mixed_combine_bwd_scalars', dout, ldd, _ptr_array(branches), len(branches), wsel, beta, x, B, Ho, Wo, C,
            ctx.reduce, dw, dbeta, scratch)
        dw = dw[:len(branches)]
        return (dout if ctx.has_prev else None, dx, dw, dbeta, None) + tuple(dbr)


class StemConvFn(torch.autograd.Function):
    """Conv2d(3, C, 3, padding=1, bias=False) on the NCHW image -> NHWC (model_search.py:111-113)."""

    @staticmethod
    def forward(ctx, image, w):
        image, w = _chk(image), _chk(w)
        B, _, H, W = image.shape
        Cout = w.shape[0]
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=image.device)
        lib('conv3x3_first_nchw', image, w, None, y, B, H, W, Cout, 0)
        ctx.save_for_backward(image, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        image, w = ctx.saved_tensors
        dy = _chk(dy)
        B, _, H, W = image.shape
        Cout = w.shape[0]
        dw = torch.empty_like(w)
        scratch = torch.empty(lib.query('conv3x3_first_wgrad_scratch_elems', B, H, W, Cout), dtype=torch.float32, device=dy.device)
        lib('conv3x3_first_wgrad', image, dy, dw, None, scratch, B, H, W, Cout)
        return None, dw


class AvgPool7FlattenFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(7) + flatten(start_dim=1) in NCHW feature order (model_search.py:130,178-180)."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x)
        B, H, W, C = x.shape
        y = torch.empty((B, C * 49), dtype=torch.float32, device=x.device)
        lib('adaptive_avgpool7_flatten', x, y, B, H, W, C)
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
        lib('adaptive_avgpool7_flatten_bwd', _chk(dy), dx, B, H, W, C)
        return dx



# ------------------------------------------------------------------------------------------ fused MixedOp edge
FUSED_EDGE = os.environ.get('LCTVQA_FUSED_EDGE', '1') != '0'     # A/B switch: 0 = one launch per primitive (the round-1 path)


def fused_edge_active():
    """The fused edge kernels are in use: the switch is on and SyncBN is not (their BatchNorm statistics are taken inside the kernels and
    cannot leave for an all-reduce; under parallel.set_sync_bn with more than one rank the per-primitive kernels run instead)."""
    return FUSED_EDGE and not sync_bn_active()


def set_fused_edge(on):
    """Use the fused edge kernels (csrc/mixed_edge.hip) where the shape allows (default) or the per-primitive kernels.
    Returns the previous setting."""
    global FUSED_EDGE
    prev, FUSED_EDGE = FUSED_EDGE, bool(on)
    return prev


EDGE_PRE_BF16 = None      # None: follow the compute dtype (bf16 storage of the pre-BN branch tensors under ops.set_compute_dtype('bf16'))


def set_edge_pre_bf16(on):
    """Storage of the nine pre-BatchNorm branch tensors of a fused edge: True = bf16 (values rounded before they are stored AND counted
    into the batch statistics; fp32 arithmetic everywhere), False = fp32, None = bf16 exactly when the compute dtype is bf16.
    Returns the previous setting."""
    global EDGE_PRE_BF16
    prev, EDGE_PRE_BF16 = EDGE_PRE_BF16, (None if on is None else bool(on))
    return prev


def edge_pre_bf16():
    return ops.linear_xent_fused() if EDGE_PRE_BF16 is None else EDGE_PRE_BF16


class EdgeWeightsFn(torch.autograd.Function):
    """The per-edge operands of a cell from its two weight tables in one autograd node: (w [E,K], b [E]) ->
    (w[0, lo:hi], b[0:1], w[1, lo:hi], b[1:2], ...) as views.  Taken one by one with ``w[e]`` / ``b[e]`` every edge costs a
    SelectBackward + SliceBackward pair (a zero-fill, a copy and an accumulation each); here the 2E small gradients are
    assembled by three launches per cell."""

    @staticmethod
    def forward(ctx, w, b, lo, hi):
        ctx.cfg = (tuple(w.shape), lo, hi)
        out = []
        for e in range(w.shape[0]):
            out += [w[e, lo:hi], b[e:e + 1]]
        return tuple(out)

    @staticmethod
    def backward(ctx, *g):
        (E, K), lo, hi = ctx.cfg
        ref = next(t for t in g if t is not None)
        gws = [t if t is not None else ref.new_zeros(hi - lo) for t in g[0::2]]
        gbs = [t if t is not None else ref.new_zeros(1) for t in g[1::2]]
        gw = ref.new_zeros((E, K))
        gw[:, lo:hi] = torch.stack(gws)
        return gw, torch.cat(gbs), None, None


_WEIGHT_GRADS = True


@contextlib.contextmanager
def weight_grads(on):
    """Inside ``weight_grads(False)`` the backward passes of the PC-DARTS ops compute input / alpha / beta gradients only and return None
    for their convolution weights: the finite-difference passes of the architects differentiate w.r.t. the architecture parameters alone
    (architect_vqa.py:75-88, architect.py:40-55), yet ``requires_grad`` of the weights -- all autograd tells a Function -- stays True."""
    global _WEIGHT_GRADS
    prev, _WEIGHT_GRADS = _WEIGHT_GRADS, bool(on)
    try:
        yield
    finally:
        _WEIGHT_GRADS = prev


class GradSlot:
    """The one gradient buffer of a state that several fused edges read.  Each consuming edge's backward adds its dx into it
    inside the kernel (``accumulate_dx``) and returns no gradient of its own; ``FanoutFn.backward`` -- which autograd runs after
    every consumer of its outputs -- hands the buffer to the state's producer.  Without it autograd sums the consumers' full-size
    dx tensors with one ATen ``add`` each (6.8 % of the darts_vqa step's kernel time)."""
    __slots__ = ('buf', 'task')

    def __init__(self):
        self.buf, self.task = None, -2

    def target(self, like):
        """(buffer, accumulate flag) for one consumer's backward in the current backward pass."""
        task = torch._C._current_graph_task_id()
        if self.buf is None or self.task != task:               # first consumer of this pass (a buffer of an unfinished pass is dropped)
            self.buf, self.task = torch.empty_like(like), task
            return self.buf, 0
        return self.buf, 1

    def take(self):
        task = torch._C._current_graph_task_id()
        buf = self.buf if self.task == task else None
        self.buf = None
        return buf


class FanoutFn(torch.autograd.Function):
    """x -> n aliases of x, one per consumer, so that no gradient of x is summed by the engine: the fused edges among the
    consumers accumulate into ``slot`` (see GradSlot), whatever else arrives (the cell's concatenation, an unfused edge) is added
    here."""

    @staticmethod
    def forward(ctx, x, n, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)                            # the fused edges return None for their alias: no zero tensors, no adds
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        buf = ctx.slot.take()
        own = buf is not None                                       # the slot's buffer is ours; an incoming gradient is not
        for g in gs:
            if g is None:
                continue
            if buf is None:
                buf = g
            elif own:
                buf.add_(g)
            else:                                                   # two ordinary gradients: never add into a tensor autograd handed us
                buf, own = buf + g, True                            # (it may be a view of the cat's grad_output or an engine buffer)
        return buf, None, None


def dense_x1(x):
    """x[..., :C/4] of an NHWC state as a dense tensor, outside autograd: made once per state that several fused edges read, so that
    their forward and backward tile loads read whole cache lines (the x1 slice is 16..64 bytes of every 64..256-byte row of x)."""
    x = _chk(x)
    B, H, W, C = x.shape
    y = torch.empty((B, H, W, C // 4), dtype=torch.float32, device=x.device)
    lib('slice_channels', x, C, 0, y, C // 4, B * H * W)
    return y


class MixedEdgeFn(torch.autograd.Function):
    """One MixedOp + its edge accumulation as ONE autograd node on the fused tile kernels:
    ``state (+)= beta * shuffle(cat(sum_k w_k bn_k(op_k(x[..., :C/4])), bypass(x[..., C/4:])))`` (model_search.py:41-55,87).
    Inputs after the fixed ones: the 12 (stride 1) or 14 (stride 2) weight tensors in the C ABI's order -- depthwise sep3.a,
    sep5.a, dil3, dil5, sep3.b, sep5.b; 1x1 of the same six; FactorizedReduce conv_1, conv_2 -- whose gradients come back as
    views of one flat buffer.  ``bns``: the BatchNorm2d holder per slot (None for the identity skip)."""

    @staticmethod
    def forward(ctx, prev, x, wsel, beta, stride, bns, training, slot, x1d, *weights):
        x, wsel, beta = _chk(x), _chk(wsel), _chk(beta)
        ctx.slot, ctx.x1d = slot, x1d
        weights = [_chk(w) for w in weights]
        B, H, W, C = x.shape
        Cq = C // 4
        Ho, Wo = H // stride, W // stride
        P, dev = B * Ho * Wo, x.device
        if prev is None:
            out = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
            acc = 0
        else:
            out, acc = prev, 1
            ctx.mark_dirty(prev)
        pbf = edge_pre_bf16() and (P * Cq) % 8 == 0                 # every slot of the bf16 buffer on a 16-byte boundary
        pre = torch.empty((9, P, Cq), dtype=torch.bfloat16 if pbf else torch.float32, device=dev)
        stats = torch.empty((2, 9, Cq), dtype=torch.float32, device=dev)          # mean, invstd per slot
        grid = lib.query('mixed_edge_grid', B, H, W, C, stride)
        part = _scratch(grid * 7 * Cq * 2 + 51 * Cq, torch.float64, dev)      # + the forward kernels' regrouped depthwise weights
        wptr = (ctypes.c_void_p * 14)(*([w.data_ptr() for w in weights] + [None] * (14 - len(weights))))
        bnp = [None] * 27
        momentum, eps = 0.1, 1e-5
        for slot, bn in enumerate(bns):
            if bn is None:
                continue
            momentum, eps = bn.momentum, bn.eps
            if bn.track_running_stats:
                bnp[slot], bnp[9 + slot], bnp[18 + slot] = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()
        lib('mixed_edge_fwd_bf16pre' if pbf else 'mixed_edge_fwd', x, B, H, W, C, stride, wptr, (ctypes.c_void_p * 27)(*bnp), pre, stats[0], stats[1], part, wsel, beta,
            out, out.stride(2), acc, int(training), float(momentum), float(eps), x1d)
        ctx.save_for_backward(x, wsel, beta, pre, stats, *weights)
        ctx.params = weights                                        # the Parameter objects (ops.grad_sink accumulates into their .grad)
        ctx.cfg = (stride, int(training), prev is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wsel, beta, pre, stats = ctx.saved_tensors[:5]
        weights = list(ctx.saved_tensors[5:])
        stride, training, has_prev = ctx.cfg
        B, H, W, C = x.shape
        Ho, Wo = H // stride, W // stride
        # the node's gradient usually arrives as a channel slice of the gradient of the cell's concatenated output: rows of
        # C floats, ldd apart -- taken as it is (the kernels take a row stride) instead of being copied
        ldd = dout.stride(2)
        if not (dout.is_cuda and dout.dtype == torch.float32 and dout.stride(3) == 1 and ldd % 4 == 0 and dout.stride(1) == Wo * ldd
                and dout.stride(0) == Ho * Wo * ldd and dout.data_ptr() % 16 == 0):
            dout = _chk(dout)
            ldd = C
        Cq, dev = C // 4, x.device
        if ctx.slot is not None:                                   # the state's shared gradient buffer (GradSlot): dx += inside the kernel
            dx, acc_dx = ctx.slot.target(x)
        else:
            dx, acc_dx = torch.empty_like(x), 0
        dw = torch.empty(7, dtype=torch.float32, device=dev)
        dbeta = torch.empty(1, dtype=torch.float32, device=dev)
        wgrad = torch.empty(lib.query('mixed_edge_wgrad_elems', C), dtype=torch.float32, device=dev)
        scratch = _scratch(lib.query('mixed_edge_bwd_scratch_floats', B, H, W, C, stride), torch.float32, dev)
        wptr = (ctypes.c_void_p * 14)(*([w.data_ptr() for w in weights] + [None] * (14 - len(weights))))
        lib('mixed_edge_bwd_bf16pre' if pre.dtype == torch.bfloat16 else 'mixed_edge_bwd', dout, ldd, x, B, H, W, C, stride, wptr, pre, stats[0], stats[1], wsel, beta, dx, dw, dbeta,
            wgrad, scratch, training, acc_dx, ctx.x1d, int(_WEIGHT_GRADS))
        grads, o = [], 0
        for w in weights:
            grads.append(wgrad[o:o + w.numel()].view_as(w) if _WEIGHT_GRADS else None)
            o += w.numel()
        if _WEIGHT_GRADS:
            # ops.grad_sink: the edge's 12-14 weight gradients go into the optimizer's gradient views with ONE multi-tensor add
            # instead of one AccumulateGrad add_ each (~730 launches per step of the search network)
            params = ctx.params
            tg = [ops.sink_target(p) for p in params]
            if tg and all(t is not None for t in tg):
                torch._foreach_add_(tg, grads)
                for p in params:
                    ops.sink_done(p)
                grads = [None] * len(grads)
        ctx.params = None
        return (dout if has_prev else None, None if ctx.slot is not None else dx, dw, dbeta, None, None, None, None, None) + tuple(grads)

# ------------------------------------------------------------------------------------------ interpreter
def _bn_args(bn):
    """(gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps) of an nn.BatchNorm2d holder."""
    training = bn.training or not bn.track_running_stats
    rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else (None, None, None)
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm2d(momentum=None) (cumulative average) in a DARTS op')
    return (bn.weight if bn.affine else None, bn.bias if bn.affine else None, rm, rv, nbt, training, bn.momentum, bn.eps)


def batchnorm(bn, x):
    """Run an nn.BatchNorm2d holder on an NHWC tensor through the HIP kernels (buffers, including num_batches_tracked,
    updated in place by the kernels)."""
    return BatchNormFn.apply(x, *_bn_args(bn))


class DwPwBnFn(torch.autograd.Function):
    """One stage of sep_conv / dil_conv (operations.py:39-63): (ReLU ->) depthwise KxK -> 1x1 -> BatchNorm2d as ONE autograd
    node: the same kernel launches as DwConvFn -> PwConvFn -> BatchNormFn, a third of their host work (the PC-DARTS path is
    host-bound at small batch)."""

    @staticmethod
    def supported(x, w_dw, w_pw):
        P = x.shape[0] * x.shape[1] * x.shape[2]            # >= the stage's output pixel count
        Cout, Cin = w_pw.shape[0], w_pw.shape[1]
        return bool(lib.query('pwconv_small_supported', P, Cout, Cin) and lib.query('pwconv_wgrad_small_supported', P, Cout, Cin))

    @staticmethod
    def forward(ctx, x, w_dw, w_pw, gamma, beta, run_mean, run_var, tracked, training, momentum, eps, stride, pad, dil, relu_in):
        x, w_dw, w_pw = _chk(x), _chk(w_dw), _chk(w_pw)
        B, H, W, C = x.shape
        K = w_dw.shape[-1]
        Ho = (H + 2 * pad - dil * (K - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (K - 1) - 1) // stride + 1
        Cout = w_pw.shape[0]
        P = B * Ho * Wo
        y1 = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
        lib('dwconv_fwd', x, w_dw, y1, B, H, W, C, K, stride, pad, dil, int(relu_in))
        y2 = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        lib('pwconv_small', y1, w_pw, None, y2, P, Cout, C, 0, 0)
        y3, mean, invstd = _bn_forward(y2, gamma, beta, run_mean, run_var, tracked, training, momentum, eps)
        ctx.save_for_backward(x, w_dw, y1, w_pw, y2, mean, invstd, gamma)
        ctx.cfg = (stride, pad, dil, int(relu_in), int(training))
        return y3

    @staticmethod
    def backward(ctx, dy):
        x, w_dw, y1, w_pw, y2, mean, invstd, gamma = ctx.saved_tensors
        stride, pad, dil, relu_in, training = ctx.cfg
        need = ctx.needs_input_grad
        B, H, W, C = x.shape
        K = w_dw.shape[-1]
        Cout = w_pw.shape[0]
        P = y2.numel() // Cout
        dev = x.device
        d2, dg, db = _bn_backward(_chk(dy), y2, mean, invstd, gamma, training)
        dw_pw = dw_dw = dx = None
        if need[2]:
            dw_pw = torch.empty_like(w_pw)
            scratch = _scratch(lib.query('pwconv_wgrad_small_scratch_elems', P, Cout, C), torch.float32, dev)
            lib('pwconv_wgrad_small', d2, y1, dw_pw, scratch, P, Cout, C, 0)
        if need[0] or need[1]:
            d1 = torch.empty_like(y1)
            lib('pwconv_small', d2, w_pw, None, d1, P, Cout, C, 0, 1)                   # dY . W
            if need[1]:
                dw_dw = torch.empty_like(w_dw)
                n = lib.query('dwconv_bwd_weight_scratch_elems', B, H, W, C, K, stride, pad, dil)
                scratch = _scratch(n, torch.float32, dev)
                lib('dwconv_bwd_weight', d1, x, dw_dw, scratch, B, H, W, C, K, stride, pad, dil, relu_in)
            if need[0]:
                dx = torch.empty_like(x)
                lib('dwconv_bwd_input', d1, w_dw, x, dx, B, H, W, C, K, stride, pad, dil, relu_in)
        return (dx, dw_dw, dw_pw, dg, db) + (None,) * 10


class PoolBnFn(torch.autograd.Function):
    """3x3 max / avg pool -> BatchNorm2d (the pooling branches of a MixedOp, model_search.py:35-37) as one autograd node."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, tracked, training, momentum, eps, stride, kind):
        x = _chk(x)
        B, H, W, C = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y1 = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
        lib('pool3x3_fwd', x, y1, B, H, W, C, stride, kind)
        y2, mean, invstd = _bn_forward(y1, gamma, beta, run_mean, run_var, tracked, training, momentum, eps)
        ctx.save_for_backward(x, y1, mean, invstd, gamma)
        ctx.cfg = (stride, kind, int(training))
        return y2

    @staticmethod
    def backward(ctx, dy):
        x, y1, mean, invstd, gamma = ctx.saved_tensors
        stride, kind, training = ctx.cfg
        B, H, W, C = x.shape
        d1, dg, db = _bn_backward(_chk(dy), y1, mean, invstd, gamma, training)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            lib('pool3x3_bwd', d1, x, dx, B, H, W, C, stride, kind)
        return (dx, dg, db) + (None,) * 8


def _is_depthwise(m):
    return isinstance(m, nn.Conv2d) and m.bias is None and m.groups == m.in_channels == m.out_channels and m.kernel_size[0] > 1


def _is_pointwise(m):
    return isinstance(m, nn.Conv2d) and m.bias is None and m.kernel_size[0] == 1 and m.groups == 1 and m.stride[0] == 1


def run_seq(seq, x):
    """Execute a reference-shaped Sequential (ReLU / Conv2d(depthwise or 1x1, bias=False) / BatchNorm2d /
    AvgPool2d(3) / MaxPool2d(3)) on an NHWC tensor.  A ReLU is fused into the convolution that follows it."""
    relu = False
    mods = list(seq)
    i = 0
    fuse_bn = not sync_bn_active()            # SyncBN: the statistics leave the kernel for an all-reduce, so BatchNorm stays a node of its own
    while i < len(mods):
        m = mods[i]
        i += 1
        if isinstance(m, nn.ReLU):
            relu = True
        elif (fuse_bn and _is_depthwise(m) and i + 1 < len(mods) and _is_pointwise(mods[i]) and isinstance(mods[i + 1], nn.BatchNorm2d)
              and DwPwBnFn.supported(x, m.weight, mods[i].weight)):
            # (ReLU ->) depthwise -> 1x1 -> BN: one autograd node
            x = DwPwBnFn.apply(x, m.weight, mods[i].weight, *_bn_args(mods[i + 1]), m.stride[0], m.padding[0], m.dilation[0], relu)
            relu = False
            i += 2
        elif fuse_bn and isinstance(m, (nn.AvgPool2d, nn.MaxPool2d)) and i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
            stride = m.stride if isinstance(m.stride, int) else m.stride[0]
            x = PoolBnFn.apply(x, *_bn_args(mods[i]), stride, 1 if isinstance(m, nn.AvgPool2d) else 0)
            i += 1
        elif isinstance(m, nn.Conv2d):
            if m.bias is not None:
                raise NotImplementedError('conv bias in a DARTS op')
            k = m.kernel_size[0]
            if k == 1 and m.groups == 1 and m.stride[0] == 1:
                x = PwConvFn.apply(x, m.weight, relu)
            elif m.groups == m.in_channels == m.out_channels:
                x = DwConvFn.apply(x, m.weight, m.stride[0], m.padding[0], m.dilation[0], relu)
            else:
                raise NotImplementedError(f'conv {m} in a DARTS op')
            relu = False
        elif isinstance(m, nn.BatchNorm2d):
            x = batchnorm(m, x)
        elif isinstance(m, nn.AvgPool2d):
            x = Pool3x3Fn.apply(x, m.stride if isinstance(m.stride, int) else m.stride[0], 1)
        elif isinstance(m, nn.MaxPool2d):
            x = Pool3x3Fn.apply(x, m.stride if isinstance(m.stride, int) else m.stride[0], 0)
        elif isinstance(m, nn.Sequential):
            x = run_seq(m, x)
        else:
            raise NotImplementedError(f'module {type(m).__name__} in a DARTS op')
    if relu:
        raise NotImplementedError('trailing ReLU')
    return x
