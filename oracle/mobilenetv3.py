"""Oracle of ``MobileNetV3Wakeword`` (src/models/architectures.py:68-123): torchvision's ``mobilenet_v3_small`` with a
one-channel stem and the reference's classifier.  torchvision (pinned 0.16.2 by the reference's requirements.txt:7) is NOT
installed here and cannot be fetched, so the body is restated from its published definition
(torchvision/models/mobilenetv3.py: ``_mobilenet_v3_conf("mobilenet_v3_small")``, ``InvertedResidual``, ``SqueezeExcitation``
with ReLU / Hardsigmoid, ``Conv2dNormActivation``, BatchNorm eps 1e-3 momentum 0.01) in plain torch.nn with the SAME module
tree, hence the same ``state_dict`` keys (``mobilenet.features.N.block.M...``, ``mobilenet.classifier.{0,3}``).
**parity unpinned** w.r.t. torchvision itself (no fixture can be generated); arithmetic = torch.nn.  Test infrastructure."""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .cnn_small import dropout_keep_mask

# input, kernel, expanded, out, use_se, activation, stride            (mobilenet_v3_small, width 1.0)
SMALL_CONF = ((16, 3, 16, 16, True, "RE", 2), (16, 3, 72, 24, False, "RE", 2), (24, 3, 88, 24, False, "RE", 1),
              (24, 5, 96, 40, True, "HS", 2), (40, 5, 240, 40, True, "HS", 1), (40, 5, 240, 40, True, "HS", 1),
              (40, 5, 120, 48, True, "HS", 1), (48, 5, 144, 48, True, "HS", 1), (48, 5, 288, 96, True, "HS", 2),
              (96, 5, 576, 96, True, "HS", 1), (96, 5, 576, 96, True, "HS", 1))
LAST_CONV, LAST_CHANNEL = 576, 1024


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def se_channels(expanded):
    return make_divisible(expanded // 4, 8)


BN = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01)


def cna(cin, cout, k, stride=1, groups=1, act=None):
    layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), BN(cout)]
    if act is not None:
        layers.append(act())
    return nn.Sequential(*layers)


class SqueezeExcitation(nn.Module):
    def __init__(self, c, cs):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1, self.fc2 = nn.Conv2d(c, cs, 1), nn.Conv2d(cs, c, 1)
        self.activation, self.scale_activation = nn.ReLU(), nn.Hardsigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return s * x


class InvertedResidual(nn.Module):
    def __init__(self, cin, k, exp, cout, use_se, act, stride):
        super().__init__()
        a = nn.Hardswish if act == "HS" else nn.ReLU
        layers = []
        if exp != cin:
            layers.append(cna(cin, exp, 1, act=a))
        layers.append(cna(exp, exp, k, stride, groups=exp, act=a))
        if use_se:
            layers.append(SqueezeExcitation(exp, se_channels(exp)))
        layers.append(cna(exp, cout, 1, act=None))
        self.block = nn.Sequential(*layers)
        self.use_res_connect = stride == 1 and cin == cout

    def forward(self, x):
        y = self.block(x)
        return x + y if self.use_res_connect else y


class _MobileNet(nn.Module):
    def __init__(self, num_classes, dropout):
        super().__init__()
        feats = [cna(1, 16, 3, 2, act=nn.Hardswish)]
        feats += [InvertedResidual(*c) for c in SMALL_CONF]
        feats.append(cna(SMALL_CONF[-1][3], LAST_CONV, 1, act=nn.Hardswish))
        self.features = nn.Sequential(*feats)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Linear(LAST_CONV, LAST_CHANNEL), nn.Hardswish(), nn.Dropout(dropout),
                                        nn.Linear(LAST_CHANNEL, num_classes))


class MobileNetV3Oracle(nn.Module):
    def __init__(self, num_classes=2, dropout=0.3, seed=0, dtype=torch.float64):
        super().__init__()
        self.mobilenet = _MobileNet(num_classes, dropout).to(dtype)
        self.p, self.seed, self.dtype = float(np.float32(dropout)), seed, dtype

    def forward(self, x, step=0, sample_offset=0, training=True):
        m = self.mobilenet
        h = m.avgpool(m.features(x.to(self.dtype))).flatten(1)
        c = m.classifier
        h = torch.nn.functional.hardswish(c[0](h))
        if training and self.p > 0:
            keep = torch.from_numpy(dropout_keep_mask(h.shape[0], h.shape[1], self.p, self.seed, step, sample_offset))
            h = h * keep.to(self.dtype) * (1.0 / (1.0 - self.p))
        return c[3](h)
