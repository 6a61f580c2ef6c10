"""bf16-native TRAINABLE ResNet trunk (BASELINE.json configs[2]: "ResNet-152+GRU bf16 ... encoder trainable"; the reference's
trainable-trunk semantics are basic_vqa/models_lct.py:56-63: no ``no_grad`` around the trunk, every layer gets a gradient).

Selected by ``ops.set_compute_dtype('bf16')`` for a trunk with trainable parameters (resnet.ResNet152Trunk.forward).  Activations and
their gradients are bf16 NHWC tensors in HBM; the fp32 master weights are converted once per forward; BatchNorm statistics, the
BatchNorm parameters' and the weights' gradients are fp32.  One autograd node per bottleneck, so the residual gradient is added
inside the data-gradient GEMM of conv1 (its epilogue addend) instead of by an ATen ``add`` per block, and the ReLU masks are applied
where the gradient is produced (the GEMM epilogue) or consumed (the BatchNorm backward sums).  Kernels: csrc/train_bf16.hip
(conv_gemm_bf16 with BatchNorm statistics in its epilogue, wgrad_bf16 on transposed LDS reads, bn_bf16_*), csrc/conv3x3_bf16.hip
(3x3 forward and, on flipped weights, data gradient), csrc/gemm_bf16.hip (stem on the matrix cores, pools)."""
import torch

from . import ops
from ._lib import lib
from .darts_ops import _scratch

from . import parallel


def set_sync_bn(on=True, process_group=None):
    """parallel.set_sync_bn (kept here for callers that only know this module)."""
    return parallel.set_sync_bn(on, process_group)


_world = parallel.sync_bn_world


def _bf16(shape, dev):
    return torch.empty(shape, dtype=torch.bfloat16, device=dev)


def _reduce_rows(part, n, C, dev, force=False):
    """fp32 partials [n][2][C] -> (buffer, is_f64, rows) the final kernels read: as they are when few, else fp64 chunk sums."""
    if n <= 32 and not force:
        return part, 0, n
    chunks = 1 if force else min(32, -(-n // 64))
    out = torch.empty((chunks, 2 * C), dtype=torch.float64, device=dev)
    lib('colreduce_f32', part, n, 2 * C, chunks, out)
    return out, 1, chunks


def bn_forward(z, part, n, bn, res, relu):
    """z bf16 [P][C] (+ the partial sums of its columns, or None: taken here) -> y = relu?(BatchNorm2d(z) (+ res)), coef [4][C]."""
    C = z.shape[-1]
    P = z.numel() // C
    dev = z.device
    training = bn.training or not bn.track_running_stats
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm2d(momentum=None)')
    coef = torch.empty((4, C), dtype=torch.float32, device=dev)
    tr = bn.track_running_stats
    rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if tr else (None, None, None)
    if training:
        if part is None:
            n = lib.query('bn_bf16_blocks', P, C)
            part = _scratch(n * 2 * C, torch.float32, dev)
            lib('bn_bf16_stats', z, P, C, part)
        world = _world()
        buf, f64, rows = _reduce_rows(part, n, C, dev, force=world > 1)
        if world > 1:
            parallel.sync_bn_all_reduce(buf)
        lib('bn_bf16_fwd_final', buf, f64, rows, C, float(P * world), float(bn.eps), float(bn.momentum), bn.weight, bn.bias, rm, rv, nbt, 1, coef)
    else:
        lib('bn_bf16_fwd_final', None, 0, 0, C, 1.0, float(bn.eps), 0.0, bn.weight, bn.bias, rm, rv, None, 0, coef)
    y = torch.empty_like(z)
    ops._timed(f'bn_apply_bf16_{C}', 0.0, 2.0 * P * C * (3 if res is not None else 2),
               lambda: lib('bn_bf16_apply', z, res, y, coef, P, C, int(relu)))
    return y, coef, training


def bn_backward(dy, ymask, z, coef, gamma, training, g_out=None, sink=(None, None)):
    """-> dz (bf16), dgamma, dbeta (fp32).  g = dy * (ymask > 0); ``g_out`` (a bf16 buffer) receives g and is then what the apply
    pass reads (the residual branch needs it as well).  ``sink``: (gamma.grad, beta.grad) tensors to accumulate into (ops.grad_sink);
    the returned gradient is None for each one given."""
    C = z.shape[-1]
    P = z.numel() // C
    dev = z.device
    n = lib.query('bn_bf16_blocks', P, C)
    part = _scratch(n * 2 * C, torch.float32, dev)
    ops._timed(f'bn_bwd_sums_bf16_{C}', 0.0, 2.0 * P * C * (2 + (ymask is not None) + (g_out is not None)),
               lambda: lib('bn_bf16_bwd_sums', dy, ymask, z, g_out, P, C, part))
    world = _world() if training else 1
    buf, f64, rows = _reduce_rows(part, n, C, dev)
    glob = None
    if world > 1:
        glob, _, _ = _reduce_rows(part, n, C, dev, force=True)
        parallel.sync_bn_all_reduce(glob)
    acc = sink[0] is not None and sink[1] is not None
    dgamma = sink[0] if acc else torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = sink[1] if acc else torch.empty(C, dtype=torch.float32, device=dev)
    bcoef = torch.empty((3, C), dtype=torch.float32, device=dev)
    lib('bn_bf16_bwd_final', buf, f64, rows, glob, C, float(P * world), coef, gamma, int(training), dgamma, dbeta, int(acc), bcoef)
    if acc:
        dgamma = dbeta = None
    dz = torch.empty_like(z)
    src, msk = (g_out, None) if g_out is not None else (dy, ymask)
    ops._timed(f'bn_bwd_apply_bf16_{C}', 0.0, 2.0 * P * C * (3 + (msk is not None)),
               lambda: lib('bn_bf16_bwd_apply', src, msk, z, dz, bcoef, P, C))
    return dz, dgamma, dbeta


# ---------------------------------------------------------------------------------------------------------- products
def _w1x1_bf16(w):
    """fp32 [Cout,Cin,1,1] -> (bf16 [Cout][Cin], bf16 [Cin][Cout]): the B operands of the forward and the data-gradient product."""
    Cout, Cin = w.shape[0], w.shape[1]
    plain = _bf16((Cout, Cin), w.device)
    trans = _bf16((Cin, Cout), w.device)
    lib('cvt_pad_bf16_both', w, Cin, Cout, Cin, plain, Cin, trans, Cout, Cout, Cin)
    return plain, trans


def conv1x1(x2, wb, stats=True, addend=None, mask=None):
    """x2 bf16 [P][K] . wb[N][K]^T -> bf16 [P][N]; stats: (partials [slabs][2][N], slabs) of the fp32 result's columns."""
    P, K = x2.shape
    N = wb.shape[0]
    dev = x2.device
    y = _bf16((P, N), dev)
    st, n = None, 0
    if stats:
        n = lib.query('conv_gemm_bf16_stats_slabs', P, N, K, 1)
        st = _scratch(n * 2 * N, torch.float32, dev)
    ops._timed(f'gemm1x1_bf16_{K}->{N}', 2.0 * P * K * N, 2.0 * P * (K + N + (N if addend is not None else 0) + (N if mask is not None else 0)),
               lambda: lib('conv_gemm_bf16', x2, K, wb, K, y, N, P, N, K, 1, 0, 0, addend, N, mask, N, st))
    return y, st, n


def wgrad(dz2, x2, Cout, Cin, taps=1, H=0, W=0, sink=None):
    """dW fp32 = dz^T . x (taps 1: [Cout,Cin]; taps 9: OIHW) from bf16 rows; ``sink``: a gradient tensor to accumulate into
    (ops.grad_sink; None is returned then)."""
    P = dz2.shape[0]
    dev = dz2.device
    need = lib.query('wgrad_bf16_workspace_bytes', P, Cout, Cin, taps, H, W)
    ws = ops._workspace(need, dev)
    dW = sink if sink is not None else torch.empty((Cout, Cin, 3, 3) if taps == 9 else (Cout, Cin, 1, 1), dtype=torch.float32, device=dev)
    ops._timed(f'wgrad_bf16_{taps}x_{Cin}->{Cout}', 2.0 * P * Cout * Cin * taps, 2.0 * P * (Cout + Cin),
               lambda: lib('wgrad_bf16', dz2, Cout, x2, Cin, P, Cout, Cin, taps, H, W, dW, int(sink is not None), ws, need))
    return None if sink is not None else dW


class _Conv3x3:
    """The 3x3 convolution of a bottleneck (stride 1, padding 1) in both directions: conv3x3_bf16's tiled kernel where it has a
    tiling for the map, else the tap loop of conv_gemm_bf16 (the 7x7 maps of the last stage)."""

    def __init__(self, w, B, H, W):
        self.Cout, self.Cin = w.shape[0], w.shape[1]
        self.B, self.H, self.W = B, H, W
        dev = w.device
        self.tiled = bool(lib.query('conv3x3_bf16_supported', B, H, W, self.Cin, self.Cout, 0)) and \
            bool(lib.query('conv3x3_bf16_supported', B, H, W, self.Cout, self.Cin, 0))
        n = w.numel()
        self.wf, self.wd = _bf16(n, dev), _bf16(n, dev)
        if self.tiled:
            lib('conv3x3_bf16_pack_weights', w, self.wf, self.Cout, self.Cin)
            lib('conv3x3_bf16_pack_weights_dgrad', w, self.wd, self.Cout, self.Cin)
        else:
            lib('pack3x3_taps_bf16', w, self.wf, self.Cout, self.Cin, 0)
            lib('pack3x3_taps_bf16', w, self.wd, self.Cout, self.Cin, 1)

    def _run(self, x, wp, Cin, Cout, stats, mask):
        B, H, W = self.B, self.H, self.W
        P = B * H * W
        dev = x.device
        y = _bf16((B, H, W, Cout), dev)
        st, n = None, 0
        if self.tiled:
            ops._timed(f'conv3x3_bf16_{H}x{W}_{Cin}->{Cout}', 2.0 * P * Cout * Cin * 9, 2.0 * P * (Cin + Cout),
                       lambda: lib('conv3x3_bf16_nhwc', x, wp, None, y, B, H, W, Cin, Cout, 0, 0))
        else:
            if stats:
                n = lib.query('conv_gemm_bf16_stats_slabs', P, Cout, Cin, 9)
                st = _scratch(n * 2 * Cout, torch.float32, dev)
            ops._timed(f'conv3x3_gemm_bf16_{H}x{W}_{Cin}->{Cout}', 2.0 * P * Cout * Cin * 9, 2.0 * P * (Cin + Cout),
                       lambda: lib('conv_gemm_bf16', x, Cin, wp, 9 * Cin, y, Cout, P, Cout, Cin, 9, H, W, None, 0, mask, Cout, st))
        return y, st, n

    def forward(self, x):
        """-> z, stats partials or None, slabs"""
        return self._run(x, self.wf, self.Cin, self.Cout, True, None)

    def dgrad(self, dz, mask):
        """-> (dx, mask_applied): the tap loop applies the ReLU mask of its output in the epilogue, the tiled kernel does not"""
        dx, _, _ = self._run(dz, self.wd, self.Cout, self.Cin, False, None if self.tiled else mask)
        return dx, (not self.tiled) and mask is not None


def _subsample(x):
    B, H, W, C = x.shape
    if H % 2 or W % 2:
        raise NotImplementedError('ResNet152Trunk: odd feature-map size at a stride-2 block')
    y = _bf16((B, H // 2, W // 2, C), x.device)
    lib('subsample2_bf16', x, y, B, H, W, C)
    return y


def _upsample_zero(d, H, W):
    B, _, _, C = d.shape
    y = _bf16((B, H, W, C), d.device)
    lib('upsample2_zero_bf16', d, y, B, H, W, C)
    return y


# ---------------------------------------------------------------------------------------------------------- autograd nodes
class StemFn(torch.autograd.Function):
    """conv1 (7x7 / stride 2) -> bn1 -> ReLU -> MaxPool(3,2,1): fp32 NCHW image -> bf16 NHWC [B, H/4, W/4, 64]."""

    @staticmethod
    def forward(ctx, image, w, gamma, beta, bn):
        image, w = ops._chk(image, 'image'), ops._chk(w)
        B, _, H, W = image.shape
        dev = image.device
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        z = _bf16((B, Ho, Wo, 64), dev)
        if lib.query('conv7x7_s2_stem_bf16_mfma_supported', B, H, W):
            lib('conv7x7_s2_stem_bf16_mfma', image, w, None, z, B, H, W, 0)
        else:
            lib('conv7x7_s2_stem_bf16', image, w, None, z, B, H, W, 0)
        y, coef, training = bn_forward(z, None, 0, bn, None, True)
        Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        out = _bf16((B, Hp, Wp, 64), dev)
        idx = torch.empty((B, Hp, Wp, 64), dtype=torch.uint8, device=dev)
        lib('maxpool3x3_s2_bf16_idx', y, out, idx, B, Ho, Wo, 64)
        ctx.save_for_backward(image, z, y, coef, gamma, idx)
        ctx.cfg = (training, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, dout):
        image, z, y, coef, gamma, idx = ctx.saved_tensors
        training, Ho, Wo = ctx.cfg
        B, _, H, W = image.shape
        dev = image.device
        dout = dout.contiguous()
        dy = _bf16((B, Ho, Wo, 64), dev)
        lib('maxpool3x3_s2_bf16_bwd', dout, idx, dy, B, Ho, Wo, 64)
        dz, dg, db = bn_backward(dy, y, z, coef, gamma, training)
        # weight gradient as a product on the matrix cores: dW[co][k] = sum_p dz[p][co] * window(p)[k] (k = (c, ky, kx) padded to 192)
        P0 = B * Ho * Wo
        cols = _bf16((P0, 192), dev)
        lib('stem_im2col_bf16', image, cols, B, H, W)
        dw = wgrad(dz.view(P0, 64), cols, 64, 192).view(64, 192)[:, :147].reshape(64, 3, 7, 7).contiguous()
        return None, dw, dg, db, None


class BottleneckFn(torch.autograd.Function):
    """torchvision's Bottleneck (stride on the 3x3) on bf16 NHWC: one autograd node.
    params = (conv1.w, bn1.w, bn1.b, conv2.w, bn2.w, bn2.b, conv3.w, bn3.w, bn3.b [, down.w, down_bn.w, down_bn.b])."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        w1, g1, b1, w2, g2, b2, w3, g3, b3 = params[:9]
        down = blk.downsample is not None
        B, H, W, Cin = x.shape
        planes = w1.shape[0]
        s = blk.stride
        x = x.contiguous()
        P = B * H * W
        w1b, w1t = _w1x1_bf16(w1)
        w3b, w3t = _w1x1_bf16(w3)
        z1, st, n = conv1x1(x.view(P, Cin), w1b)
        y1, c1, tr1 = bn_forward(z1, st, n, blk.bn1, None, True)
        c3x3 = _Conv3x3(w2, B, H, W)
        z2, st, n = c3x3.forward(y1.view(B, H, W, planes))
        if s == 2:
            z2, st, n = _subsample(z2), None, 0
        Ho, Wo = H // s, W // s
        Po = B * Ho * Wo
        z2 = z2.view(Po, planes)
        y2, c2, tr2 = bn_forward(z2, st, n, blk.bn2, None, True)
        src = zd = cd = wdt = None
        trd = False
        if down:
            wd, gd, bd = params[9:12]
            wdb, wdt = _w1x1_bf16(wd)
            src = (x if s == 1 else _subsample(x)).view(Po, Cin)
            zd, std, nd = conv1x1(src, wdb)
            idn, cd, trd = bn_forward(zd, std, nd, blk.downsample[1], None, False)
        else:
            idn = x.view(Po, Cin)
        z3, st, n = conv1x1(y2, w3b)                                    # (the statistics scratch is consumed by the next line)
        out, c3, tr3 = bn_forward(z3, st, n, blk.bn3, idn, True)
        out = out.view(B, Ho, Wo, 4 * planes)
        ctx.save_for_backward(x, z1, y1, c1, z2, y2, c2, z3, c3, out, g1, g2, g3, w1t, w3t,
                              *((src, zd, cd, params[10], wdt) if down else ()))
        ctx.c3x3 = c3x3
        ctx.blk = blk
        ctx.cfg = (down, s, planes, (tr1, tr2, tr3, trd))
        return out

    @staticmethod
    def backward(ctx, dout):
        sv = ctx.saved_tensors
        x, z1, y1, c1, z2, y2, c2, z3, c3, out, g1, g2, g3, w1t, w3t = sv[:15]
        down, s, planes, (tr1, tr2, tr3, trd) = ctx.cfg
        c3x3 = ctx.c3x3
        B, H, W, Cin = x.shape
        Ho, Wo = H // s, W // s
        P, Po = B * H * W, B * Ho * Wo
        C4 = 4 * planes
        dev = x.device
        dout = dout.contiguous().view(Po, C4)
        blk = ctx.blk
        prm = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
               blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
        if down:
            prm += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
        tg = [ops.sink_target(p) for p in prm]                          # p.grad to accumulate into, or None (ops.grad_sink)
        # bn3 (+ residual, ReLU): g = dout * (out > 0) is written once: it is the residual gradient too
        gres = _bf16((Po, C4), dev)
        dz3, dg3, db3 = bn_backward(dout, out.view(Po, C4), z3, c3, g3, tr3, g_out=gres, sink=(tg[7], tg[8]))
        dw3 = wgrad(dz3, y2, C4, planes, sink=tg[6])
        gy2, _, _ = conv1x1(dz3, w3t, stats=False, mask=y2)             # ReLU mask of y2 in the epilogue
        dz2, dg2, db2 = bn_backward(gy2, None, z2, c2, g2, tr2, sink=(tg[4], tg[5]))
        dz2f = dz2.view(B, Ho, Wo, planes) if s == 1 else _upsample_zero(dz2.view(B, Ho, Wo, planes), H, W)
        dw2 = wgrad(dz2f.view(P, planes), y1, planes, planes, 9, H, W, sink=tg[3])
        dy1, masked = c3x3.dgrad(dz2f, y1.view(B, H, W, planes))
        dz1, dg1, db1 = bn_backward(dy1.view(P, planes), None if masked else y1, z1, c1, g1, tr1, sink=(tg[1], tg[2]))
        dw1 = wgrad(dz1, x.view(P, Cin), planes, Cin, sink=tg[0])
        grads = [dw1, dg1, db1, dw2, dg2, db2, dw3, dg3, db3]
        if not down:
            dx, _, _ = conv1x1(dz1, w1t, stats=False, addend=gres)     # residual gradient added in the epilogue
        else:
            src, zd, cd, gd, wdt = sv[15:20]
            dzd, dgd, dbd = bn_backward(gres, None, zd, cd, gd, trd, sink=(tg[10], tg[11]))
            dwd = wgrad(dzd, src, C4, Cin, sink=tg[9])
            if s == 1:
                dx, _, _ = conv1x1(dz1, w1t, stats=False)
                lib('conv_gemm_bf16', dzd, C4, wdt, C4, dx, Cin, P, Cin, C4, 1, 0, 0, dx, Cin, None, 0, None)   # dx += dzd . Wd
            else:
                dsrc, _, _ = conv1x1(dzd, wdt, stats=False)
                up = _upsample_zero(dsrc.view(B, Ho, Wo, Cin), H, W)
                dx, _, _ = conv1x1(dz1, w1t, stats=False, addend=up.view(P, Cin))
            grads += [dwd, dgd, dbd]
        ctx.c3x3 = ctx.blk = None
        for p, g in zip(prm, grads):
            if g is None:                                               # accumulated in place: what AccumulateGrad would do next
                ops.sink_done(p)
        return (dx.view(B, H, W, Cin), None) + tuple(grads)


class GlobalAvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) + flatten: bf16 [B,H,W,C] -> fp32 [B,C]."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        y = torch.empty((B, C), dtype=torch.float32, device=x.device)
        lib('global_avgpool_bf16', x.contiguous(), y, B, H * W, C)
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dx = _bf16((B, H, W, C), dy.device)
        lib('global_avgpool_bf16_bwd', ops._chk(dy), dx, B, H * W, C)
        return dx


def supported(trunk, image):
    """The bf16-native path takes torchvision-shaped trunks on images whose maps stay even down to the last stage."""
    return image.dim() == 4 and image.shape[1] == 3 and image.shape[2] % 32 == 0 and image.shape[3] % 32 == 0


def forward_trainable(trunk, image):
    """ResNet152Trunk.forward for a trainable trunk under the bf16 compute dtype: [B,3,H,W] fp32 -> [B,2048] fp32."""
    x = StemFn.apply(image, trunk.conv1.weight, trunk.bn1.weight, trunk.bn1.bias, trunk.bn1)
    for layer in (trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4):
        for blk in layer:
            ps = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
                  blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
            if blk.downsample is not None:
                ps += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
            x = BottleneckFn.apply(x, blk, *ps)
    return GlobalAvgPoolFn.apply(x)
