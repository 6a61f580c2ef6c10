"""autograd.Function wrappers of the bottleneck pieces a TRAINABLE ResNet trunk needs beyond the shared ops (NHWC fp32):
BatchNorm fused with the ReLU / residual add that follows it (any channel count), the 3x3 convolution (stride 1; its backward
= the VGG trunk's wgrad / dgrad kernels, small maps as shifted GEMMs), the 7x7 / stride-2 stem and the global average pool.
1x1 convolutions are ``ops.linear`` on the rows, MaxPool(3,2,1) is ``darts_ops.Pool3x3Fn``, stride 2 is ``Subsample2Fn``."""
import torch

from . import ops, parallel
from ._lib import lib
from .ops import _chk
from .darts_ops import _scratch


class BnActFn(torch.autograd.Function):
    """y = relu?(BatchNorm2d(x) (+ residual)) on NHWC x; ``bn`` is the nn.BatchNorm2d holder (its buffers are updated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu):
        x = _chk(x)
        C = x.shape[-1]
        P = x.numel() // C
        dev = x.device
        training = bn.training or not bn.track_running_stats
        if bn.momentum is None:
            raise NotImplementedError('BatchNorm2d(momentum=None)')
        y = torch.empty_like(x)
        stats = torch.empty((2, C), dtype=torch.float32, device=dev)
        part = _scratch(lib.query('bn_wide_blocks', P) * C * 2, torch.float64, dev)
        res = _chk(residual) if residual is not None else None
        tr = bn.track_running_stats
        world = parallel.sync_bn_world() if training else 1
        if world > 1:                                               # SyncBN: pool the per-channel sums of all ranks (parallel.set_sync_bn)
            sums = torch.empty((C, 2), dtype=torch.float64, device=dev)
            lib('bn_wide_sums', x, None, None, None, None, P, C, 0, part, sums)
            parallel.sync_bn_all_reduce(sums)
            lib('bn_wide_fwd_from_sums', x, sums, float(P * world), gamma, beta, bn.running_mean if tr else None, bn.running_var if tr else None,
                bn.num_batches_tracked if tr else None, res, int(relu), y, stats[0], stats[1], P, C, float(bn.momentum), float(bn.eps))
        else:
            lib('bn_wide_fwd', x, gamma, beta, bn.running_mean if tr else None, bn.running_var if tr else None,
                bn.num_batches_tracked if tr else None, res, int(relu), y, stats[0], stats[1], part, P, C, int(training),
                float(bn.momentum), float(bn.eps))
        ctx.save_for_backward(x, y if relu else None, stats, gamma)
        ctx.cfg = (int(training), residual is not None, world)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, gamma = ctx.saved_tensors
        training, has_res, world = ctx.cfg
        dy = _chk(dy)
        C = x.shape[-1]
        P = x.numel() // C
        dev = x.device
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        sums = _scratch(2 * C, torch.float32, dev)
        part = _scratch(lib.query('bn_wide_blocks', P) * C * 2, torch.float64, dev)
        if world > 1:
            loc = torch.empty((C, 2), dtype=torch.float64, device=dev)
            lib('bn_wide_sums', x, dy, y, stats[0], stats[1], P, C, 1, part, loc)
            glob = parallel.sync_bn_all_reduce(loc.clone())
            lib('bn_wide_bwd_from_sums', dy, y, x, stats[0], stats[1], gamma, loc, glob, float(P * world), dx, dres, dg, db, sums, P, C)
        else:
            lib('bn_wide_bwd', dy, y, x, stats[0], stats[1], gamma, dx, dres, dg, db, sums, part, P, C, training)
        return dx, dg, db, dres, None, None


def _dgrad_weights(w):
    """W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]: the weights of the convolution that maps dZ to dX (layout glue)."""
    return w.detach().flip(2, 3).transpose(0, 1).contiguous()


class Conv3x3Fn(torch.autograd.Function):
    """Conv2d(Cin, Cout, 3, stride 1, padding 1, bias=False) on NHWC."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _chk(x), _chk(w)
        B, H, W, _ = x.shape
        y = ops._conv_block(x, w, None, B, H, W, 0, relu=0)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = _chk(dz)
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        dev = x.device
        dx = dw = None
        small = ops._small_map(B, H, W, Cin, Cout)
        if ctx.needs_input_grad[1]:
            if small:
                dw = _wgrad_small_map(x, dz, w)
            else:
                dw = ops.conv3x3_weight_grad(x, dz, B, H, W, Cin, Cout)
        if ctx.needs_input_grad[0]:
            if ops._small_map(B, H, W, Cout, Cin):
                dx = ops._conv3x3_small_map(dz, _dgrad_weights(w), None, B, H, W, relu=0)
            else:
                dx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
                wv = ops._wino_variant(B, H, W, Cout, Cin, 0)
                if wv is not None:
                    up = torch.empty(Cout * Cin * (36 if wv == 'wino4' else 16), dtype=torch.float32, device=dev)
                    lib(f'conv3x3_{wv}_pack_weights', w, up, Cin, Cout, 1)
                    lib(f'conv3x3_{wv}_nhwc', dz, up, None, dx, B, H, W, Cout, Cin, 0, 0)
                elif lib.query('conv3x3_uses_mfma', B, H, W, Cout, Cin, 0):
                    wp = torch.empty(Cout * Cin * 9, dtype=torch.float32, device=dev)
                    lib('conv3x3_pack_weights_dgrad', w, wp, Cout, Cin)
                    lib('conv3x3_nhwc', dz, w, wp, None, dx, None, B, H, W, Cout, Cin, 0, 0)
                else:
                    lib('conv3x3_nhwc', dz, _dgrad_weights(w), None, None, dx, None, B, H, W, Cout, Cin, 0, 0)
        return dx, dw


def _wgrad_small_map(x, dz, w):
    """dW of a 3x3 convolution on a map too small for the tiled wgrad kernel: on the zero-bordered (H+2) x (W+2) grid a tap is
    a row shift, so dW[:, :, ky, kx] = dZgrid^T . Xgrid[shifted] is one fp32 MFMA GEMM per tap (K = all grid rows)."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    dev = x.device
    Hg, Wg = H + 2, W + 2
    rows, slack = B * Hg * Wg, Wg + 1
    xg = torch.zeros((rows + 2 * slack, Cin), dtype=torch.float32, device=dev)
    zg = torch.zeros((rows, Cout), dtype=torch.float32, device=dev)
    lib('copy_window_nhwc', x, xg[slack:], B, H, W, Hg, Wg, Cin, 0, 0, 1, 1, H, W, 0)
    lib('copy_window_nhwc', dz, zg, B, H, W, Hg, Wg, Cout, 0, 0, 1, 1, H, W, 0)
    dwt = torch.empty((3, 3, Cout, Cin), dtype=torch.float32, device=dev)
    for t in range(9):
        shift = (t // 3 - 1) * Wg + (t % 3 - 1)
        a = xg[slack + shift: slack + shift + rows]
        ops.gemm(zg, 1, Cout, a, 1, Cin, Cout, Cin, rows, out=dwt[t // 3, t % 3])     # [Cout, Cin] = zg^T . a
    return dwt.permute(2, 3, 0, 1).contiguous()


class StemFn(torch.autograd.Function):
    """Conv2d(3, 64, 7, stride 2, padding 3, bias=False): fp32 NCHW image -> NHWC; gradient for the weight only."""

    @staticmethod
    def forward(ctx, image, w):
        image, w = _chk(image, 'image'), _chk(w)
        B, _, H, W = image.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, Ho, Wo, 64), dtype=torch.float32, device=image.device)
        lib('conv7x7_s2_stem', image, w, None, y, B, H, W, 0)
        ctx.save_for_backward(image)
        return y

    @staticmethod
    def backward(ctx, dz):
        (image,) = ctx.saved_tensors
        dz = _chk(dz)
        B, _, H, W = image.shape
        nb = lib.query('conv7x7_s2_stem_wgrad_blocks', B, H)
        part = torch.empty((nb, 147 * 64), dtype=torch.float32, device=image.device)
        lib('conv7x7_s2_stem_wgrad', image, dz, part, B, H, W)
        dw = ops.colsum(part).reshape(147, 64).t().contiguous().reshape(64, 3, 7, 7)
        return None, dw


class GlobalAvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) + flatten on NHWC: [B,H,W,C] -> [B,C]."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x)
        B, H, W, C = x.shape
        y = torch.empty((B, C), dtype=torch.float32, device=x.device)
        lib('global_avgpool_nhwc', x, y, B, H * W, C)
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
        lib('global_avgpool_nhwc_bwd', _chk(dy), dx, B, H * W, C)
        return dx
