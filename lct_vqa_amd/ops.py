"""Host-side op layer: thin torch.autograd.Function wrappers over the C ABI (include/lctvqa.h).

PyTorch is plumbing here (device memory, streams, autograd graph bookkeeping); every FLOP of
the hot path runs in the HIP library.  There is NO eager/CPU fallback: tensors must live on a
ROCm device and the library must be built, otherwise these functions raise.
"""
import contextlib
import os
import weakref
import threading

import torch

from ._lib import lib

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


# ----------------------------------------------------------------------------- helpers
def _chk(t, name='tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'lct_vqa_amd: {name} must be on a ROCm device (no CPU fallback)')
    if t.dtype != torch.float32:
        raise RuntimeError(f'lct_vqa_amd: {name} must be float32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty(max(nbytes // 4 + 1, 1 << 20), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def gemm(A, a_ms, a_ks, B, b_ns, b_ks, M, N, K, bias=None, addend=None, act=ACT_NONE,
         mask=None, mask_scale=1.0, out=None):
    """C[M,N] = epi(A (M x K, strides a_ms/a_ks) . B^T (N x K, strides b_ns/b_ks))."""
    dev = A.device
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dev)
    need = lib.query('gemm_workspace_bytes', M, N, K)
    ws = _workspace(need, dev) if need else None
    lib('gemm_f32', A, a_ms, a_ks, B, b_ns, b_ks, C, C.stride(0), M, N, K, bias, addend,
        addend.stride(0) if addend is not None else 0, act, mask,
        mask.stride(0) if mask is not None else 0, float(mask_scale), ws, need)
    return C


def linear_fwd(x, W, b=None, act=ACT_NONE, mask=None, mask_scale=1.0, addend=None, out=None, cache_w=False):
    """y = epi(x . W^T).  ``cache_w``: W belongs to a trunk that is frozen by construction (see _packed_weights): its bf16 copy may
    be kept between calls under the bf16 compute dtype."""
    M, K = x.shape
    N = W.shape[0]
    if _compute_dtype == 'bf16' and _bf16_able(x, W):
        Kp, Np = _ceil(K, 32), _bf16_np(N)
        xa = _bf16_rows(x, M, Kp)
        wb = _packed_weights(W, 'bf16_rows', Np * Kp, lambda dst: lib('cvt_pad_bf16', W, W.stride(0), N, K, dst, Kp, Np, Kp, 0),
                             cache_w, torch.bfloat16).view(Np, Kp)
        return _gemm_bf16x(xa, wb, M, N, Np, Kp, b, addend, act, mask, mask_scale, out)
    return gemm(x, x.stride(0), 1, W, W.stride(0), 1, M, N, K, b, addend, act, mask, mask_scale, out)


def bf16_grad_copies(dY):
    """(dY as bf16 [Mp][Np], dY^T as bf16 [Np][Mp]) from ONE pass over a Linear's output gradient (pads to multiples of 64), the
    A operands of its two backward products; (None, None) outside the bf16 compute dtype.  Hand them to ``linear_bwd_input`` /
    ``linear_bwd_weight`` when the same dY feeds several products (LSTM: dx, dW_ih, dW_hh)."""
    if _compute_dtype != 'bf16' or not (dY.is_cuda and dY.dtype == torch.float32 and dY.dim() == 2 and dY.stride(1) == 1):
        return None, None
    M, N = dY.shape
    Mp, Np = _ceil(M, 64), _ceil(N, 64)
    plain = torch.empty((Mp, Np), dtype=torch.bfloat16, device=dY.device)
    trans = torch.empty((Np, Mp), dtype=torch.bfloat16, device=dY.device)
    lib('cvt_pad_bf16_both', dY, dY.stride(0), M, N, plain, Np, trans, Mp, Mp, Np)
    return plain, trans


def linear_bwd_input(dY, W, dY_bf16=None):
    """dX[M,K] = dY[M,N] . W[N,K]"""
    M, N = dY.shape
    K = W.shape[1]
    if _compute_dtype == 'bf16' and _bf16_able(dY, W):
        Nk = _ceil(N, 64)                                      # the contraction runs over N here
        Kp = _ceil(K, 128) if K > 64 else 64
        a = dY_bf16 if dY_bf16 is not None else _bf16_rows(dY, M, Nk)
        return _gemm_bf16x(a, _bf16_rows(W, Nk, Kp, transpose=True), M, K, Kp, Nk)
    return gemm(dY, dY.stride(0), 1, W, 1, W.stride(0), M, K, N)


def linear_bwd_weight(dY, X, dYT_bf16=None):
    """dW[N,K] = dY[M,N]^T . X[M,K]"""
    M, N = dY.shape
    K = X.shape[1]
    if _compute_dtype == 'bf16' and _bf16_able(dY, X):
        Mk = _ceil(M, 64)                                      # the contraction runs over the rows
        Kp = _ceil(K, 128) if K > 64 else 64
        a = dYT_bf16 if dYT_bf16 is not None else _bf16_rows(dY, Mk, _ceil(N, 64), transpose=True)
        return _gemm_bf16x(a, _bf16_rows(X, Mk, Kp, transpose=True), N, K, Kp, Mk)
    return gemm(dY, 1, dY.stride(0), X, 1, X.stride(0), N, K, M)


# ---- bf16 compute dtype for the heads: bf16 copies of the operands (fp32 master weights, converted per call), fp32 accumulation ----
def _ceil(a, b):
    return -(-a // b) * b


def _bf16_np(N):
    """Row count of the bf16 B operand: the kernel's 128-column tile where N allows it, else 64."""
    return _ceil(N, 128) if N > 64 else 64


def _bf16_able(a, b):
    return (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
            and a.shape[0] >= 1 and a.stride(1) == 1 and b.stride(1) == 1)


def _bf16_rows(t, rows_pad, cols_pad, transpose=False):
    """Zero-padded bf16 copy of a 2-D fp32 tensor [R, C] (unit column stride): [rows_pad, cols_pad], or its transpose
    [cols_pad, rows_pad] (both pads multiples of 64 then)."""
    R, C = t.shape
    if transpose:
        dst = torch.empty((cols_pad, rows_pad), dtype=torch.bfloat16, device=t.device)
        lib('cvt_pad_bf16', t, t.stride(0), R, C, dst, rows_pad, rows_pad, cols_pad, 1)
    else:
        dst = torch.empty((rows_pad, cols_pad), dtype=torch.bfloat16, device=t.device)
        lib('cvt_pad_bf16', t, t.stride(0), R, C, dst, cols_pad, rows_pad, cols_pad, 0)
    return dst


def _gemm_bf16x(A, Bm, M, N, Np, K, bias=None, addend=None, act=ACT_NONE, mask=None, mask_scale=1.0, out=None):
    """C[M,N] fp32 = epi(A[M,K] . Bm[Np,K]^T) on the bf16 matrix cores (csrc/gemm_bf16.hip: gemm_bf16x_kernel)."""
    dev = A.device
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dev)
    need = lib.query('gemm_bf16x_workspace_bytes', M, N, Np, K)
    ws = _workspace(need, dev) if need else None
    lib('gemm_bf16x', A, A.stride(0), Bm, Bm.stride(0), C, C.stride(0), M, N, Np, K, bias, addend,
        addend.stride(0) if addend is not None else 0, act, mask, mask.stride(0) if mask is not None else 0, float(mask_scale), ws, need)
    return C


def colsum(X):
    M, N = X.shape
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    scratch = torch.empty(lib.query('colsum_scratch_elems', M, N), dtype=torch.float32, device=X.device)
    lib('colsum_f32', X, X.stride(0), M, N, out, scratch)
    return out


# ----------------------------------------------------------------------------- dropout masks
class _MaskState(threading.local):
    def __init__(self):
        self.injected = None
        self.seed = None
        self.offset = 0
        self.disabled = False


_mask_state = _MaskState()


@contextlib.contextmanager
def inject_masks(masks):
    """Parity tests: use these keep-masks (uint8/bool/float 0-1 tensors, in call order) instead of
    the device generator, exactly like the oracle's ``masks`` argument."""
    prev = _mask_state.injected
    _mask_state.injected = list(masks)
    try:
        yield
    finally:
        _mask_state.injected = prev


@contextlib.contextmanager
def dropout_disabled():
    """Make every nn.Dropout on the path the identity (parity runs of the architect steps, whose freshly built
    ``new()`` models have their Dropout holders in training mode)."""
    prev = _mask_state.disabled
    _mask_state.disabled = True
    try:
        yield
    finally:
        _mask_state.disabled = prev


def dropout_active(holder_training, p):
    return bool(holder_training) and p > 0 and not _mask_state.disabled


def manual_seed(seed):
    _mask_state.seed = int(seed)
    _mask_state.offset = 0


def dropout_mask(shape, device, p=0.5):
    """uint8 keep-mask for nn.Dropout(p) in train mode."""
    if _mask_state.injected is not None:
        m = _mask_state.injected.pop(0)
        assert tuple(m.shape) == tuple(shape), (tuple(m.shape), tuple(shape))
        return m.to(device=device, dtype=torch.uint8).contiguous()
    if _mask_state.seed is None:
        _mask_state.seed = torch.initial_seed() & 0xFFFFFFFFFFFF
    n = 1
    for s in shape:
        n *= s
    m = torch.empty(shape, dtype=torch.uint8, device=device)
    lib('dropout_mask', m, n, _mask_state.seed, _mask_state.offset, float(p))
    _mask_state.offset += n
    return m


# ----------------------------------------------------------------------------- autograd Functions
class LinearFn(torch.autograd.Function):
    """nn.Linear (+ fused act + dropout mask) -- reference basic_vqa/models.py:37,66,99-102."""

    @staticmethod
    def forward(ctx, x, W, b, act, mask, mask_scale):
        shp = x.shape
        x2 = _chk(x, 'x').reshape(-1, shp[-1])
        W = _chk(W, 'weight')
        y = linear_fwd(x2, W, b, act, mask, mask_scale)
        ctx.act, ctx.mask_scale, ctx.shp = act, mask_scale, shp
        ctx.save_for_backward(x2, W, y if (act or mask is not None) else None, mask)
        ctx.has_b = b is not None
        return y.reshape(*shp[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W, y, mask = ctx.saved_tensors
        dy = _chk(dy, 'grad').reshape(-1, W.shape[0])
        if ctx.act or mask is not None:
            dpre = torch.empty_like(dy)
            lib('act_bwd', dy, y, mask, float(ctx.mask_scale), dpre, dy.numel(), ctx.act)
        else:
            dpre = dy
        dx = dW = db = None
        g16, g16t = bf16_grad_copies(dpre) if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1]) else (None, None)
        if ctx.needs_input_grad[0]:
            dx = linear_bwd_input(dpre, W, g16).reshape(ctx.shp)
        if ctx.needs_input_grad[1]:
            dW = linear_bwd_weight(dpre, x2, g16t)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = colsum(dpre)
        return dx, dW, db, None, None, None


def linear(x, W, b=None, act=ACT_NONE, mask=None, mask_scale=1.0):
    return LinearFn.apply(x, W, b, act, mask, mask_scale)


class EmbedFn(torch.autograd.Function):
    """nn.Embedding -> (tanh) -> transpose(0,1): reference basic_vqa/models.py:58-60."""

    @staticmethod
    def forward(ctx, idx, weight, apply_tanh):
        weight = _chk(weight, 'embedding')
        if idx.dtype != torch.int64 or not idx.is_cuda:
            raise RuntimeError('lct_vqa_amd: question indices must be int64 on the device')
        idx = idx.contiguous()
        B, T = idx.shape
        V, E = weight.shape
        out = torch.empty((T, B, E), dtype=torch.float32, device=weight.device)
        lib('embed_gather', idx, weight, out, B, T, E, V, int(apply_tanh))
        ctx.save_for_backward(idx, out)
        ctx.apply_tanh, ctx.V = int(apply_tanh), V
        return out

    @staticmethod
    def backward(ctx, dy):
        idx, out = ctx.saved_tensors
        B, T = idx.shape
        E = out.shape[2]
        dy = _chk(dy, 'grad')
        demb = torch.zeros((ctx.V, E), dtype=torch.float32, device=dy.device)
        lib('embed_scatter_add', idx, dy, out, demb, B, T, E, ctx.V, ctx.apply_tanh)
        return None, demb, None


def embed(idx, weight, apply_tanh=True):
    return EmbedFn.apply(idx, weight, apply_tanh)


def _lstm_scratch(B, H, dev):
    """Scratch for the packed W_hh of one LSTM sequence call (fresh per call: the weights may have changed)."""
    return torch.empty(lib.query('lstm_seq_scratch_elems', H), dtype=torch.float32, device=dev)


class LSTMFn(torch.autograd.Function):
    """Single-layer nn.LSTM over a [T,B,I] sequence (reference basic_vqa/models.py:61,
    models_lct.py:107-108).  Returns (out [T,B,H], h_T [B,H], c_T [B,H])."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, c0):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'x')
        T, B, I = x.shape
        H = w_hh.shape[1]
        dev = x.device
        w_ih, w_hh = _chk(w_ih), _chk(w_hh)
        xp = linear_fwd(x.reshape(T * B, I), w_ih, b_ih).reshape(T, B, 4 * H)
        gates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
        h_all = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)
        c_all = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)
        if h0 is None:
            h_all[0].zero_()
            c_all[0].zero_()
        else:
            h_all[0].copy_(h0.reshape(B, H))
            c_all[0].copy_(c0.reshape(B, H))
        if lib.query('lstm_step_fused_supported', B, H):
            # the whole recurrence in one C call: W_hh packed once, one fused launch per time step
            lib('lstm_seq_fwd_bf16' if _compute_dtype == 'bf16' else 'lstm_seq_fwd', xp, w_hh, b_hh, h_all, c_all, gates, _lstm_scratch(B, H, dev), T, B, H)
        else:
            hp = torch.empty((B, 4 * H), dtype=torch.float32, device=dev)
            for t in range(T):
                linear_fwd(h_all[t], w_hh, b_hh, out=hp)
                lib('lstm_cell_fwd', xp[t], hp, c_all[t], gates[t], c_all[t + 1], h_all[t + 1], B, H)
        ctx.save_for_backward(x, w_ih, w_hh, gates, h_all, c_all)
        ctx.has_init = h0 is not None
        ctx.bf16 = _compute_dtype == 'bf16'                       # the backward pass multiplies in the dtype the forward pass used
        return h_all[1:], h_all[T].clone(), c_all[T].clone()

    @staticmethod
    def backward(ctx, dout, dhT, dcT):
        x, w_ih, w_hh, gates, h_all, c_all = ctx.saved_tensors
        T, B, I = x.shape
        H = w_hh.shape[1]
        dev = x.device
        dout = _chk(dout) if dout is not None else None
        dh_rec = _chk(dhT) if dhT is not None else None
        dc = _chk(dcT) if dcT is not None else None
        dgates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
        if lib.query('lstm_step_fused_supported', B, H):
            # step T-1 is a plain cell backward; steps T-2..0 (dh_rec = dgates[t+1] . W_hh fused with the cell backward
            # of step t) run in one C call on W_hh packed once
            dc_buf = torch.empty((2, B, H), dtype=torch.float32, device=dev)
            lib('lstm_cell_bwd', dh_rec, dout[T - 1] if dout is not None else None, dc, gates[T - 1],
                c_all[T - 1], c_all[T], dgates[T - 1], dc_buf[0], B, H)
            lib('lstm_seq_bwd_bf16' if ctx.bf16 else 'lstm_seq_bwd', dgates, w_hh, dout, gates, c_all, dc_buf, _lstm_scratch(B, H, dev), T, B, H)
            dc = dc_buf[(T - 1) & 1]
            if ctx.has_init:
                dh_rec = linear_bwd_input(dgates[0], w_hh)
        else:
            for t in range(T - 1, -1, -1):
                dc_prev = torch.empty((B, H), dtype=torch.float32, device=dev)
                lib('lstm_cell_bwd', dh_rec, dout[t] if dout is not None else None, dc, gates[t],
                    c_all[t], c_all[t + 1], dgates[t], dc_prev, B, H)
                dc = dc_prev
                if t > 0 or ctx.has_init:
                    dh_rec = linear_bwd_input(dgates[t], w_hh)
        dg2 = dgates.reshape(T * B, 4 * H)
        ni = ctx.needs_input_grad
        g16, g16t = bf16_grad_copies(dg2)                            # one conversion for the three products
        dx = linear_bwd_input(dg2, w_ih, g16).reshape(T, B, I) if ni[0] else None
        dw_ih = linear_bwd_weight(dg2, x.reshape(T * B, I), g16t) if ni[1] else None
        dw_hh = linear_bwd_weight(dg2, h_all[:T].reshape(T * B, H), g16t) if ni[2] else None
        db = colsum(dg2) if (ni[3] or ni[4]) else None
        dh0 = dh_rec if (ctx.has_init and ni[5]) else None
        dc0 = dc if (ctx.has_init and ni[6]) else None
        return dx, dw_ih, dw_hh, db, (db.clone() if db is not None else None), dh0, dc0


def lstm(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None):
    return LSTMFn.apply(x, w_ih, w_hh, b_ih, b_hh, h0, c0)


def lstm_decode_greedy(emb_weight, w_ih, w_hh, b_ih, b_hh, fc_weight, fc_bias, h0, c0, length, start_token=2):
    """Greedy question generation of the LCT examiner (reference models_lct.py:124-157) in ONE C call: returns int64 token ids
    [B, length].  x_0 = tanh(emb[start]), x_t = emb[pred_{t-1}] (no tanh), (h, c) start at (h0, c0) [B,H]; pred_t =
    argmax(fc(tanh(h_t))).  No autograd: token ids carry no gradient (the reference's ``generate`` output is integer too)."""
    with torch.no_grad():
        emb_weight, w_ih, w_hh, fc_weight = _chk(emb_weight), _chk(w_ih), _chk(w_hh), _chk(fc_weight)
        B, H = h0.shape[-2], h0.shape[-1]
        QV, We = emb_weight.shape
        dev = h0.device
        h = _chk(h0).reshape(B, H).clone()
        c = _chk(c0).reshape(B, H).clone()
        tokens = torch.empty((B, length), dtype=torch.long, device=dev)
        scratch = torch.empty(lib.query('lstm_decode_scratch_elems', B, H, We, QV), dtype=torch.float32, device=dev)
        lib('lstm_decode_greedy', emb_weight, w_ih, w_hh, b_ih, b_hh, fc_weight, fc_bias, h, c, tokens, int(start_token), int(length),
            B, H, We, QV, scratch)
    return tokens


class GRUFn(torch.autograd.Function):
    """Single-layer nn.GRU over [T,B,I] (gate order r,z,n): the GRU question-encoder variant named by BASELINE.json
    (the reference itself only has nn.LSTM).  Returns (out [T,B,H], h_T [B,H])."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0):
        ctx.set_materialize_grads(False)
        x, w_ih, w_hh = _chk(x), _chk(w_ih), _chk(w_hh)
        T, B, I = x.shape
        H = w_hh.shape[1]
        dev = x.device
        xp = linear_fwd(x.reshape(T * B, I), w_ih, b_ih).reshape(T, B, 3 * H)
        saved = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
        h_all = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)
        if h0 is None:
            h_all[0].zero_()
        else:
            h_all[0].copy_(h0.reshape(B, H))
        hp = torch.empty((B, 3 * H), dtype=torch.float32, device=dev)
        for t in range(T):
            linear_fwd(h_all[t], w_hh, b_hh, out=hp)
            lib('gru_cell_fwd', xp[t], hp, h_all[t], saved[t], h_all[t + 1], B, H)
        ctx.save_for_backward(x, w_ih, w_hh, saved, h_all)
        ctx.has_init = h0 is not None
        return h_all[1:], h_all[T].clone()

    @staticmethod
    def backward(ctx, dout, dhT):
        x, w_ih, w_hh, saved, h_all = ctx.saved_tensors
        T, B, I = x.shape
        H = w_hh.shape[1]
        dev = x.device
        dout = _chk(dout) if dout is not None else None
        dh = _chk(dhT) if dhT is not None else None
        dxp = torch.empty((T, B, 3 * H), dtype=torch.float32, device=dev)
        dhp = torch.empty((T, B, 3 * H), dtype=torch.float32, device=dev)
        for t in range(T - 1, -1, -1):
            carry = torch.empty((B, H), dtype=torch.float32, device=dev)
            lib('gru_cell_bwd', dh, dout[t] if dout is not None else None, saved[t], h_all[t], dxp[t], dhp[t], carry, B, H)
            if t > 0 or ctx.has_init:
                dh = gemm(dhp[t], 3 * H, 1, w_hh, 1, H, B, H, 3 * H, addend=carry)     # dhp . W_hh + dh*z
        ni = ctx.needs_input_grad
        dxp2, dhp2 = dxp.reshape(T * B, 3 * H), dhp.reshape(T * B, 3 * H)
        dx = linear_bwd_input(dxp2, w_ih).reshape(T, B, I) if ni[0] else None
        dw_ih = linear_bwd_weight(dxp2, x.reshape(T * B, I)) if ni[1] else None
        dw_hh = linear_bwd_weight(dhp2, h_all[:T].reshape(T * B, H)) if ni[2] else None
        db_ih = colsum(dxp2) if ni[3] else None
        db_hh = colsum(dhp2) if ni[4] else None
        dh0 = dh if (ctx.has_init and ni[5]) else None
        return dx, dw_ih, dw_hh, db_ih, db_hh, dh0


def gru(x, w_ih, w_hh, b_ih, b_hh, h0=None):
    return GRUFn.apply(x, w_ih, w_hh, b_ih, b_hh, h0)


class CatTanhFn(torch.autograd.Function):
    """tanh(reshape(transpose(cat(h,c,2)))) for num_layers = 1 (models.py:62-65)."""

    @staticmethod
    def forward(ctx, h, c):
        h, c = _chk(h), _chk(c)
        B, H = h.shape
        out = torch.empty((B, 2 * H), dtype=torch.float32, device=h.device)
        lib('cat_tanh_fwd', h, c, out, B, H)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dy):
        (out,) = ctx.saved_tensors
        B, H2 = out.shape
        dy = _chk(dy)
        dh = torch.empty((B, H2 // 2), dtype=torch.float32, device=dy.device)
        dc = torch.empty_like(dh)
        lib('cat_tanh_bwd', dy, out, dh, dc, B, H2 // 2)
        return dh, dc


class TanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x)
        y = torch.empty_like(x)
        lib('tanh_fwd', x, y, x.numel())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _chk(dy)
        dx = torch.empty_like(dy)
        lib('act_bwd', dy, y, None, 1.0, dx, dy.numel(), ACT_TANH)
        return dx


class L2NormFn(torch.autograd.Function):
    """x / ||x||_2.detach()  (reference basic_vqa/models.py:39-40): backward = dy / ||x||."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x)
        M, N = x.shape
        y = torch.empty_like(x)
        inv = torch.empty(M, dtype=torch.float32, device=x.device)
        lib('l2norm_rows_fwd', x, y, inv, M, N)
        ctx.save_for_backward(inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        (inv,) = ctx.saved_tensors
        dy = _chk(dy)
        dx = torch.empty_like(dy)
        lib('scale_rows', dy, inv, dx, dy.shape[0], dy.shape[1])
        return dx


class MulTanhDropFn(torch.autograd.Function):
    """dropout(tanh(a*b)) (reference basic_vqa/models.py:96-98)."""

    @staticmethod
    def forward(ctx, a, b, mask, scale):
        a, b = _chk(a), _chk(b)
        t = torch.empty_like(a)
        z = torch.empty_like(a)
        lib('mul_tanh_drop_fwd', a, b, mask, float(scale), t, z, a.numel())
        ctx.save_for_backward(a, b, t, mask)
        ctx.scale = scale
        return z

    @staticmethod
    def backward(ctx, dz):
        a, b, t, mask = ctx.saved_tensors
        dz = _chk(dz)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        lib('mul_tanh_drop_bwd', dz, a, b, t, mask, float(ctx.scale), da, db, a.numel())
        return da, db, None, None


class FusionFc1Fn(torch.autograd.Function):
    """Fusion head, first half (reference basic_vqa/models.py:96-100) as ONE GEMM launch:
    h1 = drop2(tanh(fc1(drop1(tanh(img * qst))))): the pointwise mul + tanh + dropout is applied to the GEMM's A
    operand while it is staged (never written to HBM).  Backward re-materialises it once for the weight gradient."""

    @staticmethod
    def forward(ctx, img, qst, W, b, mask1, scale1, mask2, scale2):
        img, qst, W = _chk(img), _chk(qst), _chk(W)
        B, E = img.shape
        A = W.shape[0]
        if _compute_dtype == 'bf16' and _bf16_able(img, W):
            # the bf16 operand copy is where mul -> tanh -> dropout is materialised (B x E values)
            t = torch.empty_like(img)
            z = torch.empty_like(img)
            lib('mul_tanh_drop_fwd', img, qst, mask1, float(scale1), t, z, img.numel())
            h1 = linear_fwd(z, W, b, ACT_TANH, mask2, float(scale2))
        else:
            h1 = torch.empty((B, A), dtype=torch.float32, device=img.device)
            need = lib.query('gemm_workspace_bytes', B, A, E)
            ws = _workspace(need, img.device) if need else None
            lib('fusion_fc1', img, qst, mask1, float(scale1), W, b, mask2, float(scale2), h1, B, E, A, ws, need)
        ctx.save_for_backward(img, qst, W, h1, mask1, mask2)
        ctx.scales = (float(scale1), float(scale2))
        return h1

    @staticmethod
    def backward(ctx, dy):
        img, qst, W, h1, mask1, mask2 = ctx.saved_tensors
        s1, s2 = ctx.scales
        dy = _chk(dy)
        dpre = torch.empty_like(dy)
        lib('act_bwd', dy, h1, mask2, s2, dpre, dy.numel(), ACT_TANH)
        t = torch.empty_like(img)
        z = torch.empty_like(img)
        lib('mul_tanh_drop_fwd', img, qst, mask1, s1, t, z, img.numel())
        g16, g16t = bf16_grad_copies(dpre)
        dW = linear_bwd_weight(dpre, z, g16t) if ctx.needs_input_grad[2] else None
        db = colsum(dpre) if ctx.needs_input_grad[3] else None
        da = dq = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dz = linear_bwd_input(dpre, W, g16)
            da = torch.empty_like(img) if ctx.needs_input_grad[0] else None
            dq = torch.empty_like(qst) if ctx.needs_input_grad[1] else None
            lib('mul_tanh_drop_bwd', dz, img, qst, t, mask1, s1, da, dq, img.numel())
        return da, dq, dW, db, None, None, None, None


class XentFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (kind 0, int64 labels) / softXEnt (kind 1, soft targets):
    reference basic_vqa/models.py:7-10,89.  Both are sum_rows / M.  softXEnt is differentiable in the
    target too (the LCT architect back-propagates through the pseudo answers, architect_lct.py:67-77)."""

    @staticmethod
    def forward(ctx, logits, target, kind):
        shp = logits.shape
        lg = _chk(logits).reshape(-1, shp[-1])
        M, N = lg.shape
        dev = lg.device
        row = torch.empty(M, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dl = torch.empty_like(lg)
        dt = None
        if kind == 0:
            labels = target.reshape(-1).contiguous()
            if labels.dtype != torch.int64:
                raise RuntimeError('labels must be int64')
            lib('xent', lg, lg.stride(0), labels, None, 0, row, loss, 1.0 / M, dl, dl.stride(0), 1.0 / M, None, M, N, 0)
        else:
            tg = _chk(target).reshape(-1, N)
            if target.requires_grad:
                dt = torch.empty_like(tg)
            lib('xent', lg, lg.stride(0), None, tg, tg.stride(0), row, loss, 1.0 / M, dl, dl.stride(0), 1.0 / M, dt, M, N, 1)
        ctx.save_for_backward(dl, dt)
        ctx.shp, ctx.tshape = shp, target.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dl, dt = ctx.saved_tensors
        g = _chk(g.reshape(1))
        out = torch.empty_like(dl)
        lib('scale_by_device_scalar', g, dl, out, dl.numel())
        gt = None
        if dt is not None:
            gt = torch.empty_like(dt)
            lib('scale_by_device_scalar', g, dt, gt, dt.numel())
            gt = gt.reshape(ctx.tshape)
        return out.reshape(ctx.shp), gt, None


class SoftmaxRowsFn(torch.autograd.Function):
    """F.softmax(x * inv_temp, dim=1), differentiable (pseudo answers: architect_lct.py:54-55,70)."""

    @staticmethod
    def forward(ctx, x, inv_temp):
        x = _chk(x)
        y = torch.empty_like(x)
        lib('softmax_rows', x, y, x.shape[0], x.shape[1], float(inv_temp))
        ctx.save_for_backward(y)
        ctx.inv_temp = float(inv_temp)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        lib('softmax_rows_bwd', _chk(dy), y, dx, y.shape[0], y.shape[1], ctx.inv_temp)
        return dx, None


class LinearXentFn(torch.autograd.Function):
    """mean_m CE(x[m] . W^T + b, labels[m]) with the logits never written to HBM: the LCT token head (reference
    basic_vqa/models_lct.py:118-120, Linear(H, QV) over every token) fused with its cross entropy (models_lct.py:253-255) on the
    bf16 matrix cores (csrc/gemm_bf16.hip: per-chunk (max, sum exp) partials in the forward GEMM's epilogue; the backward GEMM
    recomputes the tile and emits dl = (softmax - onehot) / M as bf16, row-major and transposed, the operands of dX and dW)."""

    @staticmethod
    def forward(ctx, x, W, b, labels):
        x, W = _chk(x, 'x'), _chk(W, 'weight')
        M, K = x.shape
        N = W.shape[0]
        dev = x.device
        if labels.dtype != torch.int64 or labels.numel() != M:
            raise RuntimeError('linear_xent: labels must be int64 [M]')
        labels = labels.contiguous()
        Kp, Np = _ceil(K, 32), _ceil(N, 128)
        xa = _bf16_rows(x, M, Kp)
        wb = _bf16_rows(W, Np, Kp)
        part = torch.empty(lib.query('linear_xent_bf16_part_elems', M, Np), dtype=torch.float32, device=dev)
        # the kernel writes picked[m] only where a column block contains labels[m]: a label outside [0, N) must not read
        # uninitialised memory -- it makes the loss NaN (torch's cross_entropy raises for it; no host sync here to do the same)
        picked = torch.full((M,), float('nan'), dtype=torch.float32, device=dev)
        lse = torch.empty(M, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        lib('linear_xent_bf16_fwd', xa, Kp, wb, Kp, b, labels, M, N, Np, Kp, part, picked, lse, loss)
        ctx.save_for_backward(x, W, b, labels, xa, wb, lse)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, W, b, labels, xa, wb, lse = ctx.saved_tensors
        M, K = x.shape
        N = W.shape[0]
        dev = x.device
        Kp, Np, Mp = _ceil(K, 32), _ceil(N, 128), _ceil(M, 256)
        g = _chk(g.reshape(1))
        dl = torch.empty((M, Np), dtype=torch.bfloat16, device=dev)
        dlT = torch.empty((Np, Mp), dtype=torch.bfloat16, device=dev)
        dbpart = torch.empty((2 * (Mp // 256), Np), dtype=torch.float32, device=dev)
        lib('linear_xent_bf16_bwd', xa, Kp, wb, Kp, b, labels, lse, g, M, N, Np, Kp, dl, Np, dlT, Mp, dbpart)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:                              # dX = dl . W: contraction over the (padded) vocabulary
            Ko = _ceil(K, 128) if K > 64 else 64
            dx = _gemm_bf16x(dl, _bf16_rows(W, Np, Ko, transpose=True), M, K, Ko, Np)
        if ctx.needs_input_grad[1]:                              # dW = dl^T . X: contraction over the (padded) rows
            Ko = _ceil(K, 128) if K > 64 else 64
            dW = _gemm_bf16x(dlT, _bf16_rows(x, Mp, Ko, transpose=True), N, K, Ko, Mp)
        if b is not None and ctx.needs_input_grad[2]:
            db = colsum(dbpart)[:N].contiguous()
        return dx, dW, db, None


def token_head_loss(tout, W, b, question):
    """CE(token logits[:, :-1], question[:, 1:]) of the LCT examiner (reference models_lct.py:253-255; <pad> = 0 is a class like any
    other) from the tanh'd LSTM outputs ``tout`` [T,B,H]: row t*B + b of the first T-1 steps predicts question[b, t+1]."""
    T, B, H = tout.shape
    rows = tout[:T - 1].reshape((T - 1) * B, H)
    labels = question[:, 1:].t().contiguous().reshape(-1)
    return linear_cross_entropy(rows, W, b, labels)


def linear_xent_fused():
    """True when the fused token head + cross entropy is in effect (bf16 compute dtype)."""
    return _compute_dtype == 'bf16'


def linear_cross_entropy(x, W, b, labels):
    """nn.CrossEntropyLoss()(x . W^T + b, labels) for x [M,K]: fused (logits never materialised) under the bf16 compute dtype,
    else the GEMM followed by the row-wise cross entropy."""
    if _compute_dtype == 'bf16':
        return LinearXentFn.apply(x, W, b, labels)
    return cross_entropy(linear(x, W, b), labels)


def cross_entropy(logits, labels):
    return XentFn.apply(logits, labels, 0)


def soft_xent(pred, target):
    return XentFn.apply(pred, target, 1)


def softmax_rows(x, inv_temp=1.0):
    """F.softmax(x * inv_temp, dim=1) -- forward only (pseudo labels: experiment.py:245)."""
    x = _chk(x.detach())
    y = torch.empty_like(x)
    lib('softmax_rows', x, y, x.shape[0], x.shape[1], float(inv_temp))
    return y


def argmax_rows(x):
    x = _chk(x.detach())
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    lib('argmax_rows', x, x.stride(0), x.shape[0], x.shape[1], out)
    return out


def eval_multi_choice(logits, choices, unk_idx, counts=None):
    """Row argmax of `logits` [B,A] against the multiple-choice answers `choices` [K,B] (int64, device).
    Returns (pred [B] int64, counts [2] int64 device tensor): counts[0] += hits with '<unk>' accepted,
    counts[1] += hits with '<unk>' rejected (reference experiment.py:314-324).  No host synchronisation."""
    x = _chk(logits.detach())
    if choices.dtype != torch.int64 or not choices.is_cuda:
        raise RuntimeError('eval_multi_choice: choices must be an int64 device tensor [K,B]')
    choices = choices.contiguous()
    K, B = choices.shape
    if B != x.shape[0]:
        raise RuntimeError(f'eval_multi_choice: {B} choice columns for {x.shape[0]} rows')
    if counts is None:
        counts = torch.zeros(2, dtype=torch.int64, device=x.device)
    pred = torch.empty(B, dtype=torch.int64, device=x.device)
    lib('eval_multi_choice', x, x.stride(0), B, x.shape[1], choices, K, int(unk_idx), pred, counts)
    return pred, counts


# ----------------------------------------------------------------------------- gradient sink
# ``loss.backward()`` hands every parameter gradient to an AccumulateGrad node, which -- when ``p.grad`` already exists, as the
# views of the optimizer's flat arena always do -- launches one ATen ``add_`` per parameter (478 per step for the trainable
# ResNet-152, ~300 for a PC-DARTS examiner).  With the sink on, the fused backward nodes (resnet_bf16.BottleneckFn / StemFn,
# darts_ops.MixedEdgeFn) let their weight-gradient kernels accumulate INTO ``p.grad`` (their reduce / finish kernels take an
# accumulate flag) and return None for that input; the parameter's post-accumulate-grad hooks (the arena's "reached" record, the
# data-parallel reducer's bucket counters) still run exactly once (sink_done).  OFF by default: it is only right under ``loss.backward()``;
# ``torch.autograd.grad`` (the architects' bilevel.grads) must see the gradients as return values, so bilevel.grads suspends it.
_grad_sink = {'on': False, 'suspended': 0}


def grad_sink(on=True):
    """Let the fused backward nodes accumulate parameter gradients straight into existing ``p.grad`` tensors (see above).
    Returns the previous setting."""
    global _hooks_fire_on_none
    prev = _grad_sink['on']
    _grad_sink['on'] = bool(on)
    if on and _hooks_fire_on_none is None:                      # probed here, outside any backward pass (see sink_done)
        with torch.enable_grad():
            _hooks_fire_on_none = _probe_hooks_fire_on_none()
    return prev


@contextlib.contextmanager
def no_grad_sink():
    """Suspend the sink (callers of torch.autograd.grad).  A process-wide counter, not a thread-local: the autograd engine runs the
    backward nodes of a ROCm graph on its own worker thread, which must see the suspension of the thread that called it (one driver
    thread per process: SURVEY.md section 8b, Threading)."""
    _grad_sink['suspended'] += 1
    try:
        yield
    finally:
        _grad_sink['suspended'] -= 1


def sink_target(p):
    """``p.grad`` when a backward node may accumulate into it directly, else None (the node then returns the gradient)."""
    if not _grad_sink['on'] or _grad_sink['suspended'] or p is None:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_cuda or g.shape != p.shape or not g.is_contiguous():
        return None
    return g


_hooks_fire_on_none = None


def _probe_hooks_fire_on_none():
    """Does this torch run a leaf's post-accumulate-grad hooks when the node feeding it returned None for it?  (torch 2.10 does:
    AccumulateGrad is still scheduled and its hooks run; older releases returned before them.)  Probed once, on the CPU."""
    class _P(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            return x * 2 + 0 * w.sum()

        @staticmethod
        def backward(ctx, g):
            return g * 2, None
    w = torch.nn.Parameter(torch.ones(2))
    w.grad = torch.zeros(2)
    x = torch.ones(2, requires_grad=True)
    fired = []
    h = w.register_post_accumulate_grad_hook(lambda q: fired.append(1))
    _P.apply(x, w).sum().backward()
    h.remove()
    return bool(fired)


def sink_done(p):
    """The gradient of ``p`` has been accumulated into ``p.grad`` by a kernel and the node returns None for it: make sure the
    parameter's post-accumulate-grad hooks (the arena's "reached" record, the data-parallel reducer's bucket counters) run exactly
    once -- by autograd itself where it runs them for a None gradient, else here."""
    if _hooks_fire_on_none is None:
        raise RuntimeError('ops.sink_done before ops.grad_sink(True)')
    if _hooks_fire_on_none:
        return
    hooks = getattr(p, '_post_accumulate_grad_hooks', None)
    if hooks:
        for h in list(hooks.values()):
            h(p)


# ----------------------------------------------------------------------------- per-launch timing (bench.py roofline)
_conv_timer = None


def conv_timer_start():
    """Record a pair of events (on the launch stream) around every conv3x3 MFMA launch until
    conv_timer_stop(); used by bench.py to price the dominant kernel against the MFMA roofline."""
    global _conv_timer
    _conv_timer = []


def conv_timer_stop():
    """-> list of dicts {name, ms, flops (direct-convolution count), bytes (input + output once)} (synchronises)."""
    global _conv_timer
    rec, _conv_timer = _conv_timer or [], None
    torch.cuda.synchronize()
    return [dict(name=n, ms=a.elapsed_time(b), flops=f, bytes=by) for (n, a, b, f, by) in rec]


# ----------------------------------------------------------------------------- VGG19 trunk
_conv_algo = os.environ.get('LCTVQA_CONV_ALGO', 'winograd')   # see set_conv_algo


_CONV_ALGOS = ('winograd', 'winograd2', 'direct')
_wino4_split = os.environ.get('LCTVQA_WINO4_SPLIT', '1') != '0'     # A/B switch of the split F(4x4,3x3) path (see _conv_block)
_SPLIT_CIN = int(os.environ.get('LCTVQA_SPLIT_CIN', '256'))        # least input channels of a layer that takes it (512: the rule until the batched GEMM
                                                                    # got its flat grid and buffer addressing; 256 adds conv4_1: 31.60 -> 31.39 ms/step)


def set_wino4_split(on=True):
    """Use the split F(4x4,3x3) path (transform / batched GEMM / transform) where it is the faster form.  Returns the previous setting."""
    global _wino4_split
    prev, _wino4_split = _wino4_split, bool(on)
    return prev


def set_conv_algo(name):
    """Trunk convolution algorithm, all fp32 on the MFMA pipe:
    'winograd'  (default) F(4x4,3x3) where at least 3/4 of the (padded) 4x4-tile grid is useful, else F(2x2,3x3), else direct;
    'winograd2' F(2x2,3x3) where the shape allows, else direct;
    'direct'    the implicit-GEMM kernel only.
    Returns the previous setting."""
    global _conv_algo
    if name not in _CONV_ALGOS:
        raise ValueError(name)
    prev, _conv_algo = _conv_algo, name
    return prev


def _wino_variant(B, H, W, Cin, Cout, pool):
    """'wino4' | 'wino' | None: the Winograd kernel the current algorithm setting selects for this shape."""
    if _conv_algo == 'winograd' and lib.query('conv3x3_wino4_supported', B, H, W, Cin, Cout, pool):
        return 'wino4'
    if _conv_algo in ('winograd', 'winograd2') and lib.query('conv3x3_wino_supported', B, H, W, Cin, Cout, pool):
        return 'wino'
    return None


def get_conv_algo():
    return _conv_algo


_compute_dtype = os.environ.get('LCTVQA_COMPUTE_DTYPE', 'f32')


def set_compute_dtype(name):
    """'f32' (default: the reference's arithmetic, the only headline path) or 'bf16' (BASELINE.json's bf16 configurations):
      * a FROZEN VGG19 / ResNet-152 trunk runs with bf16 activations and weights, fp32 accumulation on v_mfma_f32_32x32x16_bf16
        (csrc/conv3x3_bf16.hip, csrc/gemm_bf16.hip);
      * every nn.Linear product of the heads -- the VGG classifier, img fc, LSTM / GRU input projections, question feature fc,
        fusion fc1 / fc2, the LCT token head -- forward and both backward products, multiplies bf16 copies of its fp32 operands
        (fp32 master weights, fp32 accumulation, fp32 results: csrc/gemm_bf16.hip gemm_bf16x_kernel), and the token head is
        fused with its cross entropy where the loss is all that is wanted (``linear_cross_entropy``);
      * the recurrent products (h . W_hh per time step), the greedy decode (it emits integers), the pointwise kernels, the
        optimizer, the trainable convolutions and the PC-DARTS edges stay fp32.
    Returns the previous setting."""
    global _compute_dtype
    if name not in ('f32', 'bf16'):
        raise ValueError(name)
    prev, _compute_dtype = _compute_dtype, name
    return prev


def get_compute_dtype():
    return _compute_dtype


def _timed(name, flops, nbytes, fn):
    if _conv_timer is None:
        fn()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    _conv_timer.append((name, e0, e1, flops, nbytes))


# Packed (kernel-layout) copies of the conv weights.  By default they are re-made on every call: parameters change behind
# autograd's back (the flat-arena optimizer's kernel, the architects' `p.data.add_` perturbations,
# architect_lct.py:142-155,194-207), so nothing derived from a parameter may outlive a call.  A trunk that is frozen BY
# CONSTRUCTION (the W model's VGG19, reference basic_vqa/models.py:35-36: run under no_grad, held by no optimizer) can
# opt in to a cache (`vgg.VGG.cache_packed_weights = True`, set by bench.py): one packed copy per (parameter, stream),
# re-made when the tensor's version counter moves (load_state_dict / copy_ / any in-place op on the parameter itself)
# or the storage changes (device move).  An entry belongs to the tensor OBJECT it was made from (weak reference): another
# model's parameter that the caching allocator later places at the same address never inherits it.  In-place writes
# through `.data` do not move the counter: do not opt in for parameters that receive them.
_pack_cache = {}


def _packed_weights(w, kind, numel, pack, cache=False, dtype=torch.float32):
    if not cache:
        dst = torch.empty(numel, dtype=dtype, device=w.device)
        pack(dst)
        return dst
    key = (kind, w.data_ptr(), tuple(w.shape), torch.cuda.current_stream(w.device).cuda_stream)
    ent = _pack_cache.get(key)
    if ent is not None and ent[2]() is not w:                 # same address, other (or dead) tensor: not this one's copy
        ent = None
    if ent is None or ent[0] != w._version:
        dst = ent[1] if ent is not None else torch.empty(numel, dtype=dtype, device=w.device)
        pack(dst)
        if len(_pack_cache) > 512:                            # entries of parameters that no longer exist
            for k in [k for k, e in _pack_cache.items() if e[2]() is None]:
                del _pack_cache[k]
        _pack_cache[key] = (w._version, dst, weakref.ref(w))
        return dst
    return ent[1]


def clear_pack_cache():
    _pack_cache.clear()


def _conv3x3_small_map(cur, w, b, B, H, W, relu=1, cache=False):
    """conv3x3 + bias + ReLU on a map too small for the tiled kernels (the 7x7 maps of a ResNet's last stage): the map is
    zero-padded into a bordered (H+2) x (W+2) grid on which a tap is a constant ROW SHIFT, so the convolution is nine fp32 MFMA
    GEMMs over all grid rows accumulating into one buffer (border rows compute garbage that is cropped away).  (H+2)(W+2)/(HW)
    times the FLOPs of the layer (1.65x at 7x7), a few hundred microseconds instead of the scalar fallback kernel.
    The tap-major copy of the weights [3][3][Cout][Cin] is re-made on every call unless ``cache`` (frozen trunks only: the same
    rules as every other packed copy, `_packed_weights`)."""
    dev = cur.device
    Cout, Cin = w.shape[0], w.shape[1]
    Hg, Wg = H + 2, W + 2
    rows, slack = B * Hg * Wg, Wg + 1
    wt = _packed_weights(w, 'tap_major', 9 * Cout * Cin,
                         lambda dst: dst.view(3, 3, Cout, Cin).copy_(w.detach().permute(2, 3, 0, 1)), cache).view(3, 3, Cout, Cin)
    grid = torch.zeros((rows + 2 * slack, Cin), dtype=torch.float32, device=dev)     # slack rows: the shifted reads of the first / last rows
    lib('copy_window_nhwc', cur, grid[slack:], B, H, W, Hg, Wg, Cin, 0, 0, 1, 1, H, W, 0)
    acc = torch.empty((rows, Cout), dtype=torch.float32, device=dev)
    for t in range(9):
        shift = (t // 3 - 1) * Wg + (t % 3 - 1)
        a = grid[slack + shift: slack + shift + rows]
        gemm(a, Cin, 1, wt[t // 3, t % 3], Cin, 1, rows, Cout, Cin, b if t == 0 else None, None if t == 0 else acc, ACT_NONE, out=acc)
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
    lib('copy_window_nhwc', acc, out, B, Hg, Wg, H, W, Cout, 1, 1, 0, 0, H, W, int(relu))
    return out


def _small_map(B, H, W, Cin, Cout):
    """True when a 3x3 convolution of this shape has no tiled kernel and runs as shifted GEMMs (_conv3x3_small_map)."""
    return (not lib.query('conv3x3_uses_mfma', B, H, W, Cin, Cout, 0) and _wino_variant(B, H, W, Cin, Cout, 0) is None
            and H * W <= 256 and Cin % 4 == 0 and Cout % 4 == 0 and Cin >= 64)


def _conv_block(cur, w, b, B, H, W, pool, cache=False, relu=1):
    """conv3x3 + bias (+ ReLU) (+ fused 2x2 max pool) on an NHWC activation -> NHWC output."""
    dev = cur.device
    Cout, Cin = w.shape[0], w.shape[1]
    wv = _wino_variant(B, H, W, Cin, Cout, pool)
    if wv == 'wino4' and _wino4_split and H * W <= 28 * 28 and Cin >= _SPLIT_CIN and Cout >= 512 and B * H * W >= 4096:
        # F(4x4,3x3) as input transform + 36 batched fp32 GEMMs + output transform (csrc/conv3x3_wino_wgrad.hip, round 4): faster than
        # the fused kernel on the 28x28 and 14x14 512 -> 512 layers (profiles/r04_ab_wino_split.txt: 2.59 vs 2.85, 0.84 vs 0.93 ms alone on
        # the chip at B = 256; inside the two-stream step 2.56 vs 2.87 with the pool, 0.76 vs 0.83), slower at 56x56; the 28x28 256 -> 512 layer lost inside the
        # step until the batched GEMM got its flat grid and buffer-addressed DMA (_SPLIT_CIN); same arithmetic (exact fp32 MFMA products, the published transform matrices), 6e-6 apart
        U = _packed_weights(w, 'wino4split', 36 * Cout * Cin, lambda dst: lib('conv3x3_wino4_split_pack_weights', w, dst, Cout, Cin), cache)
        n = lib.query('conv3x3_wino4_split_workspace_elems', B, H, W, Cin, Cout)
        ws = _workspace(4 * n, dev)
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
        _timed(f'wino4_{H}x{W}_{Cin}->{Cout}{"+pool" if pool else ""}_split', 2.0 * B * H * W * Cout * Cin * 9,
               4.0 * B * (H * W * Cin + Ho * Wo * Cout),
               lambda: lib('conv3x3_wino4_split_nhwc', cur, U, b, out, ws, ws.numel(), B, H, W, Cin, Cout, int(relu), pool))
        return out
    if wv is not None:
        up = _packed_weights(w, wv, Cout * Cin * (36 if wv == 'wino4' else 16),
                             lambda dst: lib(f'conv3x3_{wv}_pack_weights', w, dst, Cout, Cin, 0), cache)
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
        _timed(f'{wv}_{H}x{W}_{Cin}->{Cout}{"+pool" if pool else ""}', 2.0 * B * H * W * Cout * Cin * 9,
               4.0 * B * (H * W * Cin + Ho * Wo * Cout),
               lambda: lib(f'conv3x3_{wv}_nhwc', cur, up, b, out, B, H, W, Cin, Cout, int(relu), pool))
        return out
    mfma = lib.query('conv3x3_uses_mfma', B, H, W, Cin, Cout, pool)
    if not mfma and not pool and H * W <= 256 and Cin % 4 == 0 and Cout % 4 == 0 and Cin >= 64:
        return _conv3x3_small_map(cur, w, b, B, H, W, relu, cache)
    wp = tmp = None
    if mfma:
        wp = _packed_weights(w, 'direct', Cout * Cin * 9, lambda dst: lib('conv3x3_pack_weights', w, dst, Cout, Cin), cache)
    elif pool:
        tmp = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
    launch = lambda: lib('conv3x3_nhwc', cur, w, wp, b, out, tmp, B, H, W, Cin, Cout, int(relu), pool)
    if mfma:
        _timed(f'conv{H}x{W}_{Cin}->{Cout}{"+pool" if pool else ""}', 2.0 * B * H * W * Cout * Cin * 9,
               4.0 * B * (H * W * Cin + Ho * Wo * Cout), launch)
    else:
        launch()
    return out


def _pool(cur, B, H, W):
    out = torch.empty((B, H // 2, W // 2, cur.shape[3]), dtype=torch.float32, device=cur.device)
    lib('maxpool2x2_nhwc', cur, out, B, H, W, cur.shape[3])
    return out


_trunk_mark = {'after': (), 'events': {}}


def set_trunk_marker(*after_convs):
    """Schedule hook for drivers that overlap the frozen trunk with other work (bench.py): every trunk forward records
    a HIP event on its stream right after launching conv layer number n for each n given (1 = conv1_1 ... 16 = conv5_4;
    none or 0 = off); ``trunk_marker_event(n)`` returns the one of the most recent forward."""
    _trunk_mark['after'] = tuple(int(a) for a in after_convs if int(a) > 0)
    _trunk_mark['events'] = {}


def trunk_marker_event(after_conv):
    return _trunk_mark['events'].get(int(after_conv))


def vgg_stem_forward(image, w, b):
    """conv1_1 + ReLU of vgg19.features: NCHW image [B,3,H,W] -> NHWC [B,H,W,64].  HBM-write bound (3.3 GB at B=256): a
    driver may run it for the next batch on a stream of its own beside the MFMA-bound layers of the current one and hand
    the result to ``vgg_features_forward(..., stem=)``."""
    image, w = _chk(image, 'image'), _chk(w)
    B, C, H, W = image.shape
    if C != 3:
        raise RuntimeError('VGG19 expects 3 input channels')
    out = torch.empty((B, H, W, w.shape[0]), dtype=torch.float32, device=image.device)
    lib('conv3x3_first_nchw', image, w, b, out, B, H, W, w.shape[0], 1)
    return out


def _bf16_plan(convs, B, H, W):
    """[(weight, bias, H, W, pool)] when every layer of this frozen trunk runs on the bf16 kernels at this input size, else None."""
    plan, n, i, first = [], len(convs), 0, True
    while i < n:
        item = convs[i]
        if isinstance(item, str):
            return None                                      # an un-fusable pool (odd map size)
        w, b = item
        pool = (i + 1 < n) and isinstance(convs[i + 1], str)
        if pool and (H % 2 or W % 2):
            return None
        if first:
            if w.shape[1] != 3 or pool:
                return None
        elif not lib.query('conv3x3_bf16_supported', B, H, W, w.shape[1], w.shape[0], int(pool)):
            return None
        plan.append((w, b, H, W, pool))
        first = False
        if pool:
            H, W = H // 2, W // 2
            i += 2
        else:
            i += 1
    return plan


def _vgg_features_forward_bf16(image, plan, cache):
    """vgg19.features + avgpool + flatten with bf16 activations (fp32 accumulation); returns fp32 [B, C*49]."""
    B, dev = image.shape[0], image.device
    cur = None
    for li, (w, b, H, W, pool) in enumerate(plan):
        w = _chk(w)
        Cout, Cin = w.shape[0], w.shape[1]
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=dev)
        if li == 0 and Cout == 64:                           # matrix cores: bf16 image and weights, fp32 accumulation
            lib('conv3x3_first_nchw_bf16_mfma', image, w, b, out, B, H, W, 1)
        elif li == 0:
            lib('conv3x3_first_nchw_bf16', image, w, b, out, B, H, W, Cout, 1)
        else:
            wp = _packed_weights(w, 'bf16', Cout * Cin * 9, lambda dst: lib('conv3x3_bf16_pack_weights', w, dst, Cout, Cin), cache,
                                 torch.bfloat16)
            _timed(f'bf16_{H}x{W}_{Cin}->{Cout}{"+pool" if pool else ""}', 2.0 * B * H * W * Cout * Cin * 9,
                   2.0 * B * (H * W * Cin + Ho * Wo * Cout), lambda: lib('conv3x3_bf16_nhwc', cur, wp, b, out, B, H, W, Cin, Cout, 1, int(pool)))
        cur = out
        if (li + 1) in _trunk_mark['after']:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            _trunk_mark['events'][li + 1] = ev
    _, Hl, Wl, Cl = cur.shape
    flat = torch.empty((B, Cl * 49), dtype=torch.float32, device=dev)
    lib('adaptive_avgpool7_flatten_bf16', cur, flat, B, Hl, Wl, Cl)
    return flat, (Hl, Wl, Cl)


def vgg_features_forward(image, convs, fuse_pool=True, keep=None, cache=False, stem=None):
    """vgg19.features + avgpool + flatten.  image NCHW [B,3,H,W]; convs: list of (weight OIHW, bias) with 'M'
    marking MaxPool2d(2,2).  When ``keep`` (a list) is given it receives, per conv layer,
    (activation after ReLU [un-pooled], H, W, pooled_after) for the backward pass; pools are then NOT
    fused into the conv epilogue because the pre-pool activation is needed."""
    image = _chk(image, 'image')
    B, C, H, W = image.shape
    if C != 3:
        raise RuntimeError('VGG19 expects 3 input channels')
    dev = image.device
    if _compute_dtype == 'bf16' and keep is None and fuse_pool and stem is None:
        plan = _bf16_plan(convs, B, H, W)                    # frozen trunk, no gradients wanted: the bf16 kernels
        if plan is not None:
            return _vgg_features_forward_bf16(image, plan, cache)
    fuse = fuse_pool and keep is None
    cur, i, n, nconv = None, 0, len(convs), 0
    while i < n:
        item = convs[i]
        if isinstance(item, str):                            # un-fused MaxPool2d(2,2)
            cur = _pool(cur, B, H, W)
            H, W = H // 2, W // 2
            i += 1
            continue
        w, b = item
        w = _chk(w)
        pool_next = (i + 1 < n) and isinstance(convs[i + 1], str)
        if cur is None:                                      # first layer: NCHW image -> NHWC
            if stem is not None:                             # precomputed by vgg_stem_forward on the same image / weights
                out = _chk(stem, 'stem')
                if tuple(out.shape) != (B, H, W, w.shape[0]):
                    raise RuntimeError(f'stem has shape {tuple(out.shape)}, expected {(B, H, W, w.shape[0])}')
            else:
                out = torch.empty((B, H, W, w.shape[0]), dtype=torch.float32, device=dev)
                lib('conv3x3_first_nchw', image, w, b, out, B, H, W, w.shape[0], 1)
            fused = 0
        else:
            fused = 1 if (fuse and pool_next and H % 2 == 0 and W % 2 == 0) else 0
            out = _conv_block(cur, w, b, B, H, W, fused, cache)
        if keep is not None:
            keep.append((out, H, W, 1 if pool_next else 0))
        cur = out
        nconv += 1
        if nconv in _trunk_mark['after']:                    # schedule hook (set_trunk_marker): this point of the stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            _trunk_mark['events'][nconv] = ev
        if fused:
            H, W = H // 2, W // 2
            i += 2
        else:
            i += 1
    Cl = cur.shape[3]
    flat = torch.empty((B, Cl * 49), dtype=torch.float32, device=dev)
    lib('adaptive_avgpool7_flatten', cur, flat, B, H, W, Cl)
    return flat, (H, W, Cl)


def vgg_trunk_forward(image, convs, fcs, train, conv_feature=None, cache=False, stem=None):
    """torchvision vgg19 (reference basic_vqa/models.py:23-28,36), forward only (frozen trunk).
    fcs: list of (weight, bias, act, drop_p) for the classifier.  Returns [B, fcs[-1] out features].
    conv_feature: the flattened features+avgpool output [B, 25088] of an earlier call on the same image and weights
    (deterministic: the dropouts live in the classifier, which is always re-run)."""
    x = conv_feature if conv_feature is not None else vgg_features_forward(image, convs, fuse_pool=True, keep=None, cache=cache,
                                                                           stem=stem)[0]
    B, dev = x.shape[0], x.device
    for (w, b, act, drop_p) in fcs:
        m = dropout_mask((B, w.shape[0]), dev, drop_p) if (train and drop_p > 0) else None
        x = linear_fwd(x, _chk(w), b, act, m, 1.0 / (1.0 - drop_p) if m is not None else 1.0, cache_w=cache)
    return x


_wgrad_algo = os.environ.get('LCTVQA_WGRAD_ALGO', 'winograd')


def set_wgrad_algo(name):
    """'winograd' (default: F(3x3,4x4) in the transformed domain where lctvqa_conv3x3_wino4_wgrad_supported says it pays, the
    direct MFMA kernel elsewhere) or 'direct'."""
    global _wgrad_algo
    if name not in ('winograd', 'direct'):
        raise ValueError(name)
    _wgrad_algo = name


def conv3x3_weight_grad(x_in, dz, B, H, W, Cin, Cout):
    """dW [Cout,Cin,3,3] of conv3x3(padding=1) from its NHWC input and the NHWC gradient of its output."""
    dev = x_in.device
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=dev)
    if _wgrad_algo == 'winograd' and lib.query('conv3x3_wino4_wgrad_supported', B, H, W, Cin, Cout):
        n = lib.query('conv3x3_wino4_wgrad_workspace_elems', B, H, W, Cin, Cout)
        ws = _workspace(4 * n, dev)
        lib('conv3x3_wino4_wgrad', x_in, dz, dw, ws, ws.numel(), B, H, W, Cin, Cout)
        return dw
    ns = lib.query('conv3x3_wgrad_scratch_elems', B, H, W, Cin, Cout)
    scratch = torch.empty(max(ns, 1), dtype=torch.float32, device=dev)
    lib('conv3x3_wgrad_nhwc', x_in, dz, dw, scratch, B, H, W, Cin, Cout)
    return dw


class VggFeaturesFn(torch.autograd.Function):
    """Trainable vgg19.features + avgpool + flatten (reference models_lct.py:62-63, pretrained=False).
    Backward per block: fused ReLU/MaxPool backward -> bias grad (column sum) -> wgrad (MFMA, pixels = K)
    -> dgrad (the forward MFMA kernel on dZ with flipped/transposed weights)."""

    @staticmethod
    def forward(ctx, image, plan, *params):
        convs, k = [], 0
        for item in plan:
            if item == 'M':
                convs.append('M')
            else:
                convs.append((params[k], params[k + 1]))
                k += 2
        keep = []
        flat, (Hl, Wl, Cl) = vgg_features_forward(image, convs, fuse_pool=False, keep=keep)
        ctx.plan, ctx.last = plan, (Hl, Wl, Cl)
        ctx.meta = [(H, W, pool) for (_, H, W, pool) in keep]
        saved = [image] + [act for (act, _, _, _) in keep]
        ctx.save_for_backward(*saved, *params)
        ctx.nl = len(keep)
        return flat

    @staticmethod
    def backward(ctx, dflat):
        saved = ctx.saved_tensors
        nl = ctx.nl
        image, acts, params = saved[0], saved[1:1 + nl], saved[1 + nl:]
        dflat = _chk(dflat)
        B, dev = image.shape[0], image.device
        Hl, Wl, Cl = ctx.last
        d = torch.empty((B, Hl, Wl, Cl), dtype=torch.float32, device=dev)
        lib('adaptive_avgpool7_flatten_bwd', dflat, d, B, Hl, Wl, Cl)
        grads = [None] * len(params)
        for li in range(nl - 1, -1, -1):
            H, W, pool = ctx.meta[li]
            act = acts[li]
            w, b = params[2 * li], params[2 * li + 1]
            Cout, Cin = w.shape[0], w.shape[1]
            odd = pool and (H % 2 or W % 2)            # floor pooling leaves the last row/col without gradient
            dz = (torch.zeros if odd else torch.empty)((B, H, W, Cout), dtype=torch.float32, device=dev)
            lib('relu_pool_bwd_nhwc', d, act, dz, B, H, W, Cout, pool)
            if li == 0:
                scratch = torch.empty(lib.query('conv3x3_first_wgrad_scratch_elems', B, H, W, Cout), dtype=torch.float32, device=dev)
                dw = torch.empty_like(w)
                db = torch.empty_like(b)
                lib('conv3x3_first_wgrad', image, dz, dw, db, scratch, B, H, W, Cout)
                grads[0], grads[1] = dw, db
                break
            x_in = acts[li - 1] if ctx.meta[li - 1][2] == 0 else None
            if x_in is None:                       # input was the pooled previous activation: recompute the pool
                Hp, Wp, _ = ctx.meta[li - 1]
                x_in = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
                lib('maxpool2x2_nhwc', acts[li - 1], x_in, B, Hp, Wp, Cin)
            grads[2 * li + 1] = colsum(dz.reshape(B * H * W, Cout))
            grads[2 * li] = conv3x3_weight_grad(x_in, dz, B, H, W, Cin, Cout)
            # dgrad: conv3x3(dz) with W'[ci][co] = flip(W[co][ci])
            dx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
            wv = _wino_variant(B, H, W, Cout, Cin, 0)
            if wv is not None:
                up = torch.empty(Cout * Cin * (36 if wv == 'wino4' else 16), dtype=torch.float32, device=dev)
                lib(f'conv3x3_{wv}_pack_weights', w, up, Cin, Cout, 1)        # dgrad conv: Cout' = Cin, Cin' = Cout
                lib(f'conv3x3_{wv}_nhwc', dz, up, None, dx, B, H, W, Cout, Cin, 0, 0)
            elif lib.query('conv3x3_uses_mfma', B, H, W, Cout, Cin, 0):
                wp = torch.empty(Cout * Cin * 9, dtype=torch.float32, device=dev)
                lib('conv3x3_pack_weights_dgrad', w, wp, Cout, Cin)
                lib('conv3x3_nhwc', dz, w, wp, None, dx, None, B, H, W, Cout, Cin, 0, 0)
            else:
                wt = w.flip(2, 3).transpose(0, 1).contiguous()       # layout glue for the direct fallback kernel
                lib('conv3x3_nhwc', dz, wt, None, None, dx, None, B, H, W, Cout, Cin, 0, 0)
            d = dx
        return (None, None) + tuple(grads)
