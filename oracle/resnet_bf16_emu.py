"""TEST INFRASTRUCTURE ONLY (checker; the product never imports this).

The ResNet trunk of oracle/resnet_oracle.py (torchvision's published Bottleneck network; PARITY UNPINNED by the reference, which has
no ResNet -- SURVEY.md section 0.1) computed in float64 with the STORAGE ROUNDINGS of the bf16 training path inserted: every
activation, activation gradient and weight operand that lct_vqa_amd/resnet_bf16.py keeps in bf16 is rounded to bf16 here at the same
point; statistics, BatchNorm coefficients and parameter gradients stay wide.  Why it exists: a ReLU whose pre-activation lies within
one bf16 rounding of zero takes the other branch than in exact arithmetic, and its whole gradient term flips with it (about 1 element in
1000 per layer, 2-4 % of a layer's gradient in L2, accumulating over depth).  That makes "bf16 path vs exact fp32 oracle" a statement
about bf16 itself; "bf16 path vs this file" pins the kernels (same decisions, tolerance = accumulation order), and
"this file vs resnet_oracle" (tests/test_oracle_golden.py) measures what bf16 storage costs."""
import torch
import torch.nn.functional as F

BF = torch.bfloat16


def r(t):
    """round to bf16, keep computing wide"""
    return t.to(torch.float32).to(BF).to(torch.float64)


def f32(t):
    return t.to(torch.float32).to(torch.float64)


def _nchw(t):
    return t.permute(0, 3, 1, 2)


def _nhwc(t):
    return t.permute(0, 2, 3, 1)


class BnState:
    """coefficients of one BatchNorm application (training: batch statistics of `zstat`)"""

    def __init__(self, zstat, bn, training):
        C = zstat.shape[-1]
        flat = zstat.reshape(-1, C)
        self.P = flat.shape[0]
        self.training = training
        if training:
            self.mean = flat.mean(0)
            self.var = flat.var(0, unbiased=False)
        else:
            self.mean, self.var = bn['running_mean'].double(), bn['running_var'].double()
        self.invstd = 1.0 / torch.sqrt(self.var + bn['eps'])
        self.gamma, self.beta = bn['weight'].double(), bn['bias'].double()
        self.mean32, self.invstd32 = f32(self.mean), f32(self.invstd)
        self.scale = f32(self.gamma * self.invstd)
        self.shift = f32(self.beta - self.mean * self.gamma * self.invstd)

    def apply(self, z, res=None, relu=True):
        v = f32(z * self.scale + self.shift)
        if res is not None:
            v = v + res
        return r(F.relu(v) if relu else v)

    def backward(self, g, z):
        """g: masked output gradient (bf16 values), z: stored (rounded) pre-BatchNorm tensor -> dz (rounded), dgamma, dbeta"""
        C = z.shape[-1]
        gf, zf = g.reshape(-1, C), z.reshape(-1, C)
        s0, sz = gf.sum(0), (gf * zf).sum(0)
        s1 = self.invstd32 * (sz - self.mean32 * s0)
        A = self.gamma * self.invstd32
        if self.training:
            Bc = f32(-A * self.invstd32 * s1 / self.P)
            Cc = f32(A * (self.mean32 * self.invstd32 * s1 - s0) / self.P)
        else:
            Bc = Cc = torch.zeros_like(A)
        return r(f32(A) * g + Bc * z + Cc), s1, s0


def _bn_cfg(sd, prefix, training_override=None):
    return {'weight': sd[prefix + '.weight'], 'bias': sd[prefix + '.bias'], 'running_mean': sd[prefix + '.running_mean'],
            'running_var': sd[prefix + '.running_var'], 'eps': 1e-5}


def conv1x1(x, wq):
    return torch.einsum('bhwc,oc->bhwo', x, wq)


def conv3x3(x, wq):
    return _nhwc(F.conv2d(_nchw(x), wq, padding=1))


def conv3x3_dgrad(dz, wq):
    return _nhwc(F.conv_transpose2d(_nchw(dz), wq, padding=1))


def conv3x3_wgrad(dz, x, Cout, Cin):
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(_nchw(x), w, padding=1).backward(_nchw(dz))
    return w.grad


def _up(d, H, W):
    B, _, _, C = d.shape
    out = torch.zeros(B, H, W, C, dtype=d.dtype)
    out[:, ::2, ::2] = d
    return out


def block(x, sd, pre, stride, down, training, tiled3x3, grads, dout_fn):
    """One bottleneck forward; returns (out, backward closure).  x: bf16-valued float64 NHWC."""
    w1q, w2q, w3q = r(sd[pre + 'conv1.weight'].double()[:, :, 0, 0]), r(sd[pre + 'conv2.weight'].double()), r(sd[pre + 'conv3.weight'].double()[:, :, 0, 0])
    planes = w1q.shape[0]
    z1f = conv1x1(x, w1q)
    bn1 = BnState(f32(z1f), _bn_cfg(sd, pre + 'bn1'), training)
    z1 = r(z1f)
    y1 = bn1.apply(z1)
    z2f = conv3x3(y1, w2q)
    if stride == 2:
        z2 = r(z2f)[:, ::2, ::2]
        bn2 = BnState(z2, _bn_cfg(sd, pre + 'bn2'), training)
    else:
        z2 = r(z2f)
        bn2 = BnState(z2 if tiled3x3 else f32(z2f), _bn_cfg(sd, pre + 'bn2'), training)
    y2 = bn2.apply(z2)
    z3f = conv1x1(y2, w3q)
    bn3 = BnState(f32(z3f), _bn_cfg(sd, pre + 'bn3'), training)
    z3 = r(z3f)
    if down:
        wdq = r(sd[pre + 'downsample.0.weight'].double()[:, :, 0, 0])
        src = x if stride == 1 else x[:, ::2, ::2]
        zdf = conv1x1(src, wdq)
        bnd = BnState(f32(zdf), _bn_cfg(sd, pre + 'downsample.1'), training)
        zd = r(zdf)
        idn = bnd.apply(zd, relu=False)
    else:
        idn = x
    out = bn3.apply(z3, res=idn)
    stats = {pre + 'bn1': bn1, pre + 'bn2': bn2, pre + 'bn3': bn3}
    if down:
        stats[pre + 'downsample.1'] = bnd

    def backward(dout):
        B, H, W, Cin = x.shape
        gres = dout * (out > 0)
        dz3, dg, db = bn3.backward(gres, z3)
        grads[pre + 'bn3.weight'], grads[pre + 'bn3.bias'] = dg, db
        grads[pre + 'conv3.weight'] = torch.einsum('bhwo,bhwc->oc', dz3, y2)[:, :, None, None]
        gy2 = r(f32(torch.einsum('bhwo,oc->bhwc', dz3, w3q)) * (y2 > 0))
        dz2, dg, db = bn2.backward(gy2, z2)
        grads[pre + 'bn2.weight'], grads[pre + 'bn2.bias'] = dg, db
        dz2f = dz2 if stride == 1 else _up(dz2, H, W)
        grads[pre + 'conv2.weight'] = conv3x3_wgrad(dz2f, y1, planes, planes)
        dy1 = f32(conv3x3_dgrad(dz2f, w2q))
        if tiled3x3:
            g1 = r(dy1) * (y1 > 0)
        else:
            g1 = r(dy1 * (y1 > 0))
        dz1, dg, db = bn1.backward(g1, z1)
        grads[pre + 'bn1.weight'], grads[pre + 'bn1.bias'] = dg, db
        grads[pre + 'conv1.weight'] = torch.einsum('bhwo,bhwc->oc', dz1, x)[:, :, None, None]
        d1 = f32(torch.einsum('bhwo,oc->bhwc', dz1, w1q))
        if not down:
            return r(d1 + gres)
        dzd, dg, db = bnd.backward(gres, zd)
        grads[pre + 'downsample.1.weight'], grads[pre + 'downsample.1.bias'] = dg, db
        grads[pre + 'downsample.0.weight'] = torch.einsum('bhwo,bhwc->oc', dzd, src)[:, :, None, None]
        dd = f32(torch.einsum('bhwo,oc->bhwc', dzd, wdq))
        if stride == 1:
            return r(dd + r(d1))
        return r(d1 + _up(r(dd), H, W))

    return out, backward, stats


def trunk_forward_backward(sd, layers, image, dy, training=True, tiled=lambda H, W, C: W % 8 == 0 or W % 14 == 0):
    """-> (features [B,2048] float64, {parameter name: gradient}, {bn name: BnState}).  sd: a resnet_oracle / ResNet152Trunk
    state_dict (fp32); image fp32 NCHW; dy [B,2048].  ``tiled(H, W, planes)``: whether the HIP path runs that map's 3x3 convolutions on the
    tiled kernel (statistics of the ROUNDED output, ReLU mask applied after the rounding) or on the tap loop."""
    grads, stats = {}, {}
    img = image.double()
    wq = r(sd['conv1.weight'].double())
    z0 = r(_nhwc(F.conv2d(r(img), wq, stride=2, padding=3)))
    bn0 = BnState(z0, _bn_cfg(sd, 'bn1'), training)
    stats['bn1'] = bn0
    y0 = bn0.apply(z0)
    y0n = _nchw(y0).clone().requires_grad_(True)
    x = F.max_pool2d(y0n, 3, 2, 1)
    pooled = x
    x = _nhwc(x.detach())
    backs = []
    inplanes = 64
    for li, nblk in enumerate(layers):
        planes = 64 * 2 ** li
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 0) else 1
            down = bi == 0
            pre = f'layer{li + 1}.{bi}.'
            x, bw, st = block(x, sd, pre, stride, down, training, tiled(x.shape[1], x.shape[2], planes), grads, None)
            stats.update(st)
            backs.append(bw)
            inplanes = planes * 4
    B, H, W, C = x.shape
    feat = f32(x.reshape(B, H * W, C).sum(1) / (H * W))
    d = r(f32(dy.double() / (H * W)))[:, None, None, :].expand(B, H, W, C)
    for bw in reversed(backs):
        d = bw(d)
    pooled.backward(_nchw(d))
    dy0 = r(f32(_nhwc(y0n.grad)))
    dz0, dg, db = bn0.backward(dy0 * (y0 > 0), z0)
    grads['bn1.weight'], grads['bn1.bias'] = dg, db
    w = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(r(img), w, stride=2, padding=3).backward(_nchw(dz0))       # the stem's weight gradient multiplies bf16 input windows
    grads['conv1.weight'] = w.grad
    return feat, grads, stats
