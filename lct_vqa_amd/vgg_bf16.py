"""bf16-native TRAINABLE VGG19 features (reference basic_vqa/models_lct.py:56-63: ``pretrained=False`` trains the examiner's trunk):
vgg19.features + avgpool + flatten as ONE autograd node on bf16 NHWC activations, selected by ``ops.set_compute_dtype('bf16')``.
Forward = the frozen bf16 trunk's kernels with the pools un-fused (the backward needs the pre-pool activations); backward per
layer: ReLU / MaxPool backward in one pass (csrc/train_bf16.hip) -> bias gradient (column sums) -> weight gradient (wgrad_bf16 on
transposed LDS reads, fp32 result) -> data gradient (conv3x3_bf16 on the flipped / transposed weights).  fp32 master weights,
fp32 parameter gradients (accumulated into ``p.grad`` directly under ops.grad_sink)."""
import torch

from . import ops
from ._lib import lib
from .darts_ops import _scratch


def plan_supported(plan_shapes, B):
    """plan_shapes: [(Cout, Cin, H, W, pool)] per conv layer.  The bf16 kernels must tile every layer in both directions."""
    for li, (Cout, Cin, H, W, pool) in enumerate(plan_shapes):
        if li == 0:
            if Cin != 3 or Cout != 64:
                return False
            continue
        if not lib.query('conv3x3_bf16_supported', B, H, W, Cin, Cout, 0) or not lib.query('conv3x3_bf16_supported', B, H, W, Cout, Cin, 0):
            return False
        if Cout % 64 or Cin % 64 or (Cout != 64 and Cout % 128) or Cout & (Cout - 1):
            return False
    return True


def _colsum_bf16(t2, C):
    """fp32 [C] column sums of a bf16 [P][C] tensor (the bias gradient)."""
    P = t2.numel() // C
    dev = t2.device
    n = lib.query('bn_bf16_blocks', P, C)
    part = _scratch(n * 2 * C, torch.float32, dev)
    lib('bn_bf16_stats', t2, P, C, part)
    out = torch.empty((1, 2 * C), dtype=torch.float64, device=dev)
    lib('colreduce_f32', part, n, 2 * C, 1, out)
    return out[0, :C].float()


class VggFeaturesBf16Fn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, image, plan, module_params, *params):
        image = ops._chk(image, 'image')
        B, _, H, W = image.shape
        dev = image.device
        acts, meta, k, cur = [], [], 0, None
        i, n = 0, len(plan)
        while i < n:
            w, b = params[k], params[k + 1]
            k += 2
            pool = (i + 1 < n) and plan[i + 1] == 'M'
            Cout, Cin = w.shape[0], w.shape[1]
            out = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
            if cur is None:
                lib('conv3x3_first_nchw_bf16_mfma', image, w, b, out, B, H, W, 1)
            else:
                wp = torch.empty(Cout * Cin * 9, dtype=torch.bfloat16, device=dev)
                lib('conv3x3_bf16_pack_weights', w, wp, Cout, Cin)
                ops._timed(f'conv3x3_bf16_{H}x{W}_{Cin}->{Cout}', 2.0 * B * H * W * Cout * Cin * 9, 2.0 * B * H * W * (Cin + Cout),
                           lambda: lib('conv3x3_bf16_nhwc', cur, wp, b, out, B, H, W, Cin, Cout, 1, 0))
            acts.append(out)
            meta.append((H, W, pool, cur is None))
            cur = out
            if pool:
                pooled = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.bfloat16, device=dev)
                lib('maxpool2x2_bf16', cur, pooled, B, H, W, Cout)
                cur, H, W = pooled, H // 2, W // 2
                i += 2
            else:
                i += 1
        Cl = cur.shape[3]
        flat = torch.empty((B, Cl * 49), dtype=torch.float32, device=dev)
        lib('adaptive_avgpool7_flatten_bf16', cur, flat, B, H, W, Cl)
        ctx.meta, ctx.last, ctx.module_params = meta, (H, W, Cl), module_params
        ctx.save_for_backward(image, *acts, *params)
        ctx.nl = len(acts)
        return flat

    @staticmethod
    def backward(ctx, dflat):
        saved = ctx.saved_tensors
        nl = ctx.nl
        image, acts, params = saved[0], saved[1:1 + nl], saved[1 + nl:]
        prm = ctx.module_params
        B, dev = image.shape[0], image.device
        Hl, Wl, Cl = ctx.last
        d = torch.empty((B, Hl, Wl, Cl), dtype=torch.bfloat16, device=dev)
        lib('adaptive_avgpool7_flatten_bwd_bf16', ops._chk(dflat), d, B, Hl, Wl, Cl)
        grads = [None] * len(params)

        def give(idx, g):
            """gradient of parameter idx: into its .grad when the sink is on, else returned"""
            tgt = ops.sink_target(prm[idx])
            if tgt is not None:
                tgt.add_(g.view_as(tgt))
                ops.sink_done(prm[idx])
            else:
                grads[idx] = g

        for li in range(nl - 1, -1, -1):
            H, W, pool, first = ctx.meta[li]
            act = acts[li]
            w = params[2 * li]
            Cout, Cin = w.shape[0], w.shape[1]
            P = B * H * W
            dz = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
            ops._timed(f'relu_pool_bwd_bf16_{Cout}', 0.0, 2.0 * P * Cout * (2 + (0.25 if pool else 1)),
                       lambda: lib('relu_pool_bwd_bf16', d, act, dz, B, H, W, Cout, int(pool)))
            if first:
                dzf = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
                lib('cvt_bf16_to_f32', dz, dzf, dz.numel())
                scratch = torch.empty(lib.query('conv3x3_first_wgrad_scratch_elems', B, H, W, Cout), dtype=torch.float32, device=dev)
                dw, db = torch.empty_like(w), torch.empty(Cout, dtype=torch.float32, device=dev)
                lib('conv3x3_first_wgrad', image, dzf, dw, db, scratch, B, H, W, Cout)
                give(0, dw)
                give(1, db)
                break
            if ctx.meta[li - 1][2]:                                 # this layer read the pooled previous activation
                Hp, Wp = ctx.meta[li - 1][0], ctx.meta[li - 1][1]
                x_in = torch.empty((B, H, W, Cin), dtype=torch.bfloat16, device=dev)
                lib('maxpool2x2_bf16', acts[li - 1], x_in, B, Hp, Wp, Cin)
            else:
                x_in = acts[li - 1]
            give(2 * li + 1, _colsum_bf16(dz.view(P, Cout), Cout))
            tgt = ops.sink_target(prm[2 * li])
            need = lib.query('wgrad_bf16_workspace_bytes', P, Cout, Cin, 9, H, W)
            ws = ops._workspace(need, dev)
            dw = tgt if tgt is not None else torch.empty_like(w)
            ops._timed(f'wgrad_bf16_9x_{Cin}->{Cout}', 2.0 * P * Cout * Cin * 9, 2.0 * P * (Cout + Cin),
                       lambda: lib('wgrad_bf16', dz, Cout, x_in, Cin, P, Cout, Cin, 9, H, W, dw, int(tgt is not None), ws, need))
            if tgt is not None:
                ops.sink_done(prm[2 * li])
            else:
                grads[2 * li] = dw
            wd = torch.empty(Cout * Cin * 9, dtype=torch.bfloat16, device=dev)
            lib('conv3x3_bf16_pack_weights_dgrad', w, wd, Cout, Cin)
            dx = torch.empty((B, H, W, Cin), dtype=torch.bfloat16, device=dev)
            ops._timed(f'conv3x3_bf16_dgrad_{H}x{W}_{Cout}->{Cin}', 2.0 * P * Cout * Cin * 9, 2.0 * P * (Cin + Cout),
                       lambda: lib('conv3x3_bf16_nhwc', dz, wd, None, dx, B, H, W, Cout, Cin, 0, 0))
            d = dx
        ctx.module_params = None
        return (None, None, None) + tuple(grads)
