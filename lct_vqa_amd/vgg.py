"""VGG19 (configuration E) module shell with torchvision's attribute names / state_dict keys
(``features.{0,2,5,...}``, ``classifier.{0,3,6}``) whose forward runs the HIP trunk.

Stands where the reference calls ``torchvision.models.vgg19`` (basic_vqa/models.py:23,
models_lct.py:46, darts_vqa/vqa_model.py:19).  ``pretrained=True`` would download ImageNet
weights; offline it is a seeded random init (torchvision's own ``_initialize_weights`` scheme).
Like the reference, callers may truncate ``classifier`` with
``nn.Sequential(*list(model.classifier.children())[:-1])`` -- forward interprets whatever
Linear/ReLU/Dropout sequence is present.
"""
import torch
import torch.nn as nn

from . import ops

CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
         512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class VGG(nn.Module):
    # opt-in: keep the kernel-layout copies of the conv weights between calls of the no-grad path (ops._packed_weights);
    # only for a trunk whose parameters nobody updates in place behind the version counter
    cache_packed_weights = False

    def __init__(self, num_classes=1000):
        super().__init__()
        layers, cin = [], 3
        for v in CFG_E:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(
            nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _plan(self):
        convs = []
        for m in self.features:
            if isinstance(m, nn.Conv2d):
                convs.append((m.weight, m.bias))
            elif isinstance(m, nn.MaxPool2d):
                convs.append('M')
        fcs, mods, i = [], list(self.classifier), 0
        while i < len(mods):
            lin = mods[i]
            if not isinstance(lin, nn.Linear):
                raise RuntimeError(f'unexpected classifier module {type(lin).__name__}')
            act, p = ops.ACT_NONE, 0.0
            i += 1
            if i < len(mods) and isinstance(mods[i], nn.ReLU):
                act = ops.ACT_RELU
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout):
                p = mods[i].p if ops.dropout_active(mods[i].training, mods[i].p) else 0.0   # the holder's own mode
                i += 1
            fcs.append((lin.weight, lin.bias, act, p))
        return convs, fcs

    def conv_features(self, x):
        """features + avgpool + flatten of a trunk that needs no gradient: [B, 512*7*7].  Deterministic (no dropout), so
        a caller that evaluates the same frozen trunk on the same batch twice may pass it back as `conv_feature`."""
        convs, _ = self._plan()
        with torch.no_grad():
            return ops.vgg_features_forward(x, convs, fuse_pool=True, keep=None, cache=self.cache_packed_weights)[0]

    def stem(self, x):
        """conv1_1 + ReLU alone (NHWC), for a driver that runs it ahead on its own stream: pass it back as ``stem``."""
        convs, _ = self._plan()
        with torch.no_grad():
            return ops.vgg_stem_forward(x, *convs[0])

    def forward(self, x, conv_feature=None, stem=None):
        convs, fcs = self._plan()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not needs_grad:
            with torch.no_grad():
                return ops.vgg_trunk_forward(x, convs, fcs, True, conv_feature, cache=self.cache_packed_weights, stem=stem)
        if conv_feature is not None or stem is not None:
            raise RuntimeError('conv_feature / stem can only be re-used with a trunk that needs no gradient')
        # trainable trunk (reference models_lct.py:62-63): autograd through the HIP backward kernels
        plan = ['M' if isinstance(c, str) else 'C' for c in convs]
        params = [t for c in convs if not isinstance(c, str) for t in c]
        h = None
        if ops.get_compute_dtype() == 'bf16' and x.dim() == 4 and x.shape[1] == 3:
            # bf16 activations end to end (vgg_bf16.py) where the bf16 kernels tile every layer in both directions
            from . import vgg_bf16
            shapes, H, W = [], x.shape[2], x.shape[3]
            for i, c in enumerate(convs):
                if isinstance(c, str):
                    H, W = H // 2, W // 2
                else:
                    shapes.append((c[0].shape[0], c[0].shape[1], H, W, i + 1 < len(convs) and isinstance(convs[i + 1], str)))
            if vgg_bf16.plan_supported(shapes, x.shape[0]):
                h = vgg_bf16.VggFeaturesBf16Fn.apply(x, plan, params, *params)
        if h is None:
            h = ops.VggFeaturesFn.apply(x, plan, *params)
        for (w, b, act, p) in fcs:
            train = p > 0
            m = ops.dropout_mask((h.shape[0], w.shape[0]), h.device, p) if train else None
            h = ops.linear(h, w, b, act, m, 1.0 / (1.0 - p) if train else 1.0)
        return h


def vgg19(pretrained=False, **kw):
    return VGG(**kw)


def truncated_vgg19(pretrained=True):
    """(VGG19 without its last classifier layer, width of what is left): the image trunk every encoder of the reference
    builds (basic_vqa/models.py:23-28, models_lct.py:46-51, darts_vqa/vqa_model.py:19-22)."""
    model = vgg19(pretrained=pretrained)
    width = model.classifier[-1].in_features
    model.classifier = nn.Sequential(*list(model.classifier.children())[:-1])
    return model, width


def embed_feature(feature, fc):
    """fc(feature) divided by its (detached) L2 norm -- the tail of every image encoder (models.py:37-40)."""
    return ops.L2NormFn.apply(ops.linear(feature, fc.weight, fc.bias))
