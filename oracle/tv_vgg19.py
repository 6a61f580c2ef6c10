"""TEST INFRASTRUCTURE -- not part of the product path.

Restatement of the *published architecture* of ``torchvision.models.vgg19``
(Simonyan & Zisserman 2014, configuration "E").  torchvision is a third-party
dependency of the reference that is NOT vendored under /root/reference and is
not installed in this image (version unpinned by the reference: it has no
requirements file; >=0.3 because the reference relies on AdaptiveAvgPool2d).
Reference call sites:  basic_vqa/models.py:23, basic_vqa/models_lct.py:46,
darts_vqa/vqa_model.py:19  (``models.vgg19(pretrained=...)``).

The reference pins no result at this boundary, so VGG19 itself is
"parity unpinned by the reference"; everything around it is pinned by running
the reference's own modules on top of this architecture (tests/make_golden.py).

Layout (module names = state_dict keys of torchvision):
  features   : Conv3x3(p=1)+ReLU stacks, cfg E =
               [64,64,M,128,128,M,256,256,256,256,M,512,512,512,512,M,512,512,512,512,M]
  avgpool    : AdaptiveAvgPool2d((7,7))
  classifier : Linear(25088,4096) ReLU Dropout Linear(4096,4096) ReLU Dropout Linear(4096,1000)
Init (torchvision ``_initialize_weights``): conv kaiming_normal_(fan_out, relu),
bias 0; linear N(0, 0.01), bias 0.  ``pretrained`` cannot be honoured offline
(weights would be downloaded) and is ignored: seeded random init.
"""
import torch
import torch.nn as nn

CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
         512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class VGG(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        layers = []
        cin = 3
        for v in CFG_E:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers.append(nn.Conv2d(cin, v, kernel_size=3, padding=1))
                layers.append(nn.ReLU(inplace=True))
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

    def forward(self, x):
        x = self.features(x)
        x = self.avgpool(x)
        x = torch.flatten(x, 1)
        x = self.classifier(x)
        return x


def vgg19(pretrained=False, **kw):
    return VGG(**kw)
