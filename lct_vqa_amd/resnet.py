"""Frozen ResNet-152 image trunk on the HIP kernels (BASELINE.json configs[2], "ResNet-152+GRU"; the reference has no ResNet,
SURVEY.md section 0.1: parity unpinned by the reference, checked against oracle/resnet_oracle.py).

``ResNet152Trunk`` holds torchvision-named parameters and buffers (``conv1``, ``bn1``, ``layer{1..4}.{i}.conv{1,2,3}`` /
``bn{1,2,3}`` / ``downsample.{0,1}``) so a torchvision ``resnet152`` state_dict loads.  Two execution modes:
  * FROZEN (no gradient wanted: under ``no_grad`` or all parameters frozen): a feature extractor with its BatchNorms on their
    running statistics whatever the module's mode; the statistics are then a per-channel affine map folded into the preceding
    convolution (csrc/resnet.hip), so every layer is one kernel with a fused bias + ReLU (+ residual) epilogue;
    with ``ops.set_compute_dtype('bf16')`` the same layers run on the bf16 kernels (bf16 activations and folded weights, fp32
    accumulation: bf16 GEMM for 1x1, conv3x3_bf16 for 3x3, fp32 for the 7x7 maps of the last stage);
  * TRAINABLE (gradients wanted): every convolution, BatchNorm (batch statistics in train mode, running statistics in eval mode)
    and pool is an autograd node on the HIP kernels (resnet_ops.py), nn.Module semantics throughout.

Execution: NHWC fp32; 7x7/s2 stem kernel; MaxPool(3,2,1) = pool3x3; 3x3 convolutions on the VGG trunk's kernels
(Winograd F(4x4,3x3) / direct MFMA, bias + ReLU fused; stride 2 = stride 1 then a 2x subsample); 1x1 convolutions as the
strided fp32 MFMA GEMM over NHWC rows with the epilogue relu(acc + shift + identity); global average pool."""
import torch
import torch.nn as nn

from . import ops
from ._lib import lib


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNet152Trunk(nn.Module):
    """torchvision resnet152 minus its fc: [B,3,H,W] -> [B,2048]."""

    def __init__(self, layers=(3, 8, 36, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.out_features = 2048
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        # id(conv) -> (versions, w', shift), kept only on request (``cache_folded_weights``, the analogue of
        # vgg.VGG.cache_packed_weights; bench.py sets it for its frozen trunk).  By default every no_grad forward re-folds:
        # what changes a trunk's tensors between two of them -- the flat-arena optimizer's kernel, bn_wide_fwd's
        # running-statistics update, the architects' lctvqa_multi_copy refills -- writes through raw pointers and never moves
        # a version counter, so a trainable trunk evaluated under no_grad (the driver's validation pass between training
        # epochs, experiment.py:150-176, 292-303) would otherwise be served the first validation's weights for ever.
        self.cache_folded_weights = False
        self._folded = {}
        self._cache_ok = False

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes, 1, None) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    # ---- BatchNorm (running statistics) folded into the convolution in front of it -----------------------------------
    def _fold(self, conv, bn):
        key = id(conv)
        ver = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               conv.weight.data_ptr())
        hit = self._folded.get(key) if self._cache_ok else None
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        w = ops._chk(conv.weight)
        Cout = w.shape[0]
        wf = torch.empty_like(w)
        shift = torch.empty(Cout, dtype=torch.float32, device=w.device)
        lib('fold_bn', w, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), Cout, w.numel() // Cout, wf, shift)
        if self._cache_ok:
            self._folded[key] = (ver, wf, shift)
        return wf, shift

    def _conv1x1(self, x, conv, bn, relu, identity=None):
        """x NHWC -> NHWC: relu?(x . w'^T + shift (+ identity))"""
        B, H, W, Cin = x.shape
        wf, shift = self._fold(conv, bn)
        Cout = wf.shape[0]
        P = B * H * W
        y = torch.empty((P, Cout), dtype=torch.float32, device=x.device)
        ops._timed(f'gemm1x1_{H}x{W}_{Cin}->{Cout}', 2.0 * P * Cin * Cout, 4.0 * P * (Cin + Cout * (1 if identity is None else 2)),
                   lambda: ops.linear_fwd(x.reshape(P, Cin), wf.reshape(Cout, Cin), shift, ops.ACT_RELU if relu else ops.ACT_NONE,
                                          addend=None if identity is None else identity.reshape(P, Cout), out=y))
        return y.reshape(B, H, W, Cout)

    def _conv3x3(self, x, conv, bn, stride):
        B, H, W, _ = x.shape
        wf, shift = self._fold(conv, bn)
        # conv + shift + ReLU at stride 1; the packed form of the folded weights is kept only with the folded weights themselves
        y = ops._conv_block(x, wf, shift, B, H, W, 0, cache=self._cache_ok)
        return y if stride == 1 else self._subsample(y)

    @staticmethod
    def _subsample(x):
        B, H, W, C = x.shape
        y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
        if H % 2 or W % 2:
            raise NotImplementedError('ResNet152Trunk: odd feature-map size at a stride-2 block')
        lib('subsample2_fwd', x, y, B, H, W, C, 0, 0)
        return y

    # ---- trainable path: nn.Module semantics, every layer an autograd node ------------------------------------------
    @staticmethod
    def _t_conv1x1(x, conv):
        B, H, W, Cin = x.shape
        Cout = conv.weight.shape[0]
        return ops.linear(x.reshape(B * H * W, Cin), conv.weight.reshape(Cout, Cin)).reshape(B, H, W, Cout)

    def _forward_trainable(self, image):
        from . import resnet_ops as R
        from . import darts_ops as D
        bn = lambda x, m, relu, res=None: R.BnActFn.apply(x, m.weight, m.bias, res, m, relu)
        x = bn(R.StemFn.apply(image, self.conv1.weight), self.bn1, True)
        x = D.Pool3x3Fn.apply(x, 2, 0)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                identity = x
                if blk.downsample is not None:
                    src = x if blk.stride == 1 else D.Subsample2Fn.apply(x, False)[0]
                    identity = bn(self._t_conv1x1(src, blk.downsample[0]), blk.downsample[1], False)
                out = bn(self._t_conv1x1(x, blk.conv1), blk.bn1, True)
                out = R.Conv3x3Fn.apply(out, blk.conv2.weight)
                if blk.stride == 2:
                    out = D.Subsample2Fn.apply(out, False)[0]
                out = bn(out, blk.bn2, True)
                x = bn(self._t_conv1x1(out, blk.conv3), blk.bn3, True, identity)
        return R.GlobalAvgPoolFn.apply(x)

    # ---- frozen path in bf16 (ops.set_compute_dtype('bf16')): bf16 activations / weights, fp32 accumulation ----------
    def _fold_bf16(self, conv, bn, packed3x3=False):
        """(bf16 weights, fp32 shift) of a folded convolution: [Cout][Cin] rows for the GEMM, or the 3x3 kernel's packed layout."""
        wf, shift = self._fold(conv, bn)
        key = ('bf16', id(conv), packed3x3)
        hit = self._folded.get(key) if self._cache_ok else None
        if hit is not None and hit[0] is wf:
            return hit[1], shift
        Cout, Cin = wf.shape[0], wf.shape[1]
        wb = torch.empty(wf.numel(), dtype=torch.bfloat16, device=wf.device)
        if packed3x3:
            lib('conv3x3_bf16_pack_weights', wf, wb, Cout, Cin)
        else:
            lib('cvt_f32_to_bf16', wf, wb, wf.numel())
        if self._cache_ok:
            self._folded[key] = (wf, wb)
        return wb, shift

    def _b_conv1x1(self, x, conv, bn, relu, identity=None):
        B, H, W, Cin = x.shape
        wb, shift = self._fold_bf16(conv, bn)
        Cout = conv.weight.shape[0]
        P = B * H * W
        y = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=x.device)
        ops._timed(f'gemm1x1_bf16_{H}x{W}_{Cin}->{Cout}', 2.0 * P * Cin * Cout, 2.0 * P * (Cin + Cout * (1 if identity is None else 2)),
                   lambda: lib('gemm_bf16', x, Cin, wb, Cin, y, Cout, P, Cout, Cin, shift, identity, Cout, int(relu), 1))
        return y

    def _b_conv3x3(self, x, conv, bn, stride):
        B, H, W, Cin = x.shape
        Cout = conv.weight.shape[0]
        dev = x.device
        if lib.query('conv3x3_bf16_supported', B, H, W, Cin, Cout, 0):
            wb, shift = self._fold_bf16(conv, bn, packed3x3=True)
            y = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
            ops._timed(f'bf16_{H}x{W}_{Cin}->{Cout}', 2.0 * B * H * W * Cout * Cin * 9, 2.0 * B * H * W * (Cin + Cout),
                       lambda: lib('conv3x3_bf16_nhwc', x, wb, shift, y, B, H, W, Cin, Cout, 1, 0))
        else:                                                    # maps without a bf16 tiling (7x7): the fp32 path of this layer
            xf = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
            lib('cvt_bf16_to_f32', x, xf, x.numel())
            yf = self._conv3x3(xf, conv, bn, 1)
            y = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
            lib('cvt_f32_to_bf16', yf, y, yf.numel())
        if stride == 1:
            return y
        if H % 2 or W % 2:
            raise NotImplementedError('ResNet152Trunk: odd feature-map size at a stride-2 block')
        z = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.bfloat16, device=dev)
        lib('subsample2_bf16', y, z, B, H, W, Cout)
        return z

    def _forward_frozen_bf16(self, image):
        image = ops._chk(image, 'image')
        B, C, H, W = image.shape
        dev = image.device
        wf, shift = self._fold(self.conv1, self.bn1)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        x = torch.empty((B, Ho, Wo, 64), dtype=torch.bfloat16, device=dev)
        if lib.query('conv7x7_s2_stem_bf16_mfma_supported', B, H, W):     # matrix cores: bf16 image and weights, fp32 accumulation
            lib('conv7x7_s2_stem_bf16_mfma', image, wf, shift, x, B, H, W, 1)
        else:
            lib('conv7x7_s2_stem_bf16', image, wf, shift, x, B, H, W, 1)
        Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        y = torch.empty((B, Hp, Wp, 64), dtype=torch.bfloat16, device=dev)
        lib('maxpool3x3_s2_bf16', x, y, B, Ho, Wo, 64)
        x = y
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                identity = x
                if blk.downsample is not None:
                    src = x
                    if blk.stride == 2:
                        Bs, Hs, Ws, Cs = x.shape
                        if Hs % 2 or Ws % 2:
                            raise NotImplementedError('ResNet152Trunk: odd feature-map size at a stride-2 block')
                        src = torch.empty((Bs, Hs // 2, Ws // 2, Cs), dtype=torch.bfloat16, device=dev)
                        lib('subsample2_bf16', x, src, Bs, Hs, Ws, Cs)
                    identity = self._b_conv1x1(src, blk.downsample[0], blk.downsample[1], relu=False)
                out = self._b_conv1x1(x, blk.conv1, blk.bn1, relu=True)
                out = self._b_conv3x3(out, blk.conv2, blk.bn2, blk.stride)
                x = self._b_conv1x1(out, blk.conv3, blk.bn3, relu=True, identity=identity)
        Bf, Hf, Wf, Cf = x.shape
        feat = torch.empty((Bf, Cf), dtype=torch.float32, device=dev)
        lib('global_avgpool_bf16', x, feat, Bf, Hf * Wf, Cf)
        return feat

    def forward(self, image):
        trainable = any(p.requires_grad for p in self.parameters())
        if torch.is_grad_enabled() and trainable:
            if ops.get_compute_dtype() == 'bf16':
                from . import resnet_bf16
                if resnet_bf16.supported(self, image):                    # bf16 activations end to end (resnet_bf16.py)
                    return resnet_bf16.forward_trainable(self, image)
            return self._forward_trainable(image)
        # a trunk WITHOUT a trainable parameter is written by nobody between two forwards (the optimizer arena, multi_copy refills
        # and the running-statistics kernels only touch trainable trunks; load_state_dict / copy_ move the version counters the
        # cache key checks): fold once.  A trainable trunk evaluated under no_grad re-folds every call (see __init__).
        self._cache_ok = bool(self.cache_folded_weights) or not trainable
        if not self._cache_ok and self._folded:
            self._folded.clear()
        if ops.get_compute_dtype() == 'bf16' and image.shape[1] == 3:
            with torch.no_grad():
                return self._forward_frozen_bf16(image)
        with torch.no_grad():
            image = ops._chk(image, 'image')
            B, C, H, W = image.shape
            if C != 3:
                raise RuntimeError('ResNet expects 3 input channels')
            wf, shift = self._fold(self.conv1, self.bn1)
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            x = torch.empty((B, Ho, Wo, 64), dtype=torch.float32, device=image.device)
            lib('conv7x7_s2_stem', image, wf, shift, x, B, H, W, 1)
            Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
            y = torch.empty((B, Hp, Wp, 64), dtype=torch.float32, device=image.device)
            lib('pool3x3_fwd', x, y, B, Ho, Wo, 64, 2, 0)
            x = y
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in layer:
                    identity = x
                    if blk.downsample is not None:
                        src = x if blk.stride == 1 else self._subsample(x)
                        identity = self._conv1x1(src, blk.downsample[0], blk.downsample[1], relu=False)
                    out = self._conv1x1(x, blk.conv1, blk.bn1, relu=True)
                    out = self._conv3x3(out, blk.conv2, blk.bn2, blk.stride)
                    x = self._conv1x1(out, blk.conv3, blk.bn3, relu=True, identity=identity)
            B, H, W, C = x.shape
            feat = torch.empty((B, C), dtype=torch.float32, device=x.device)
            lib('global_avgpool_nhwc', x, feat, B, H * W, C)
            return feat
