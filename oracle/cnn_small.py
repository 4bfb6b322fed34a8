"""``cnn_small`` in plain ``torch.nn`` -- the model oracle.

The reference has NO ``cnn_small`` (``src/models/architectures.py:458-509`` lists
resnet18/mobilenetv3/lstm/gru/tcn only).  BASELINE.json's configs name
it, so the build defines it (DS-CNN-S style): stem
``Conv2d(1,64,3,s2,p1,bias=False)+BN+ReLU`` (the same stem form the reference puts in
front of MobileNetV3, ``architectures.py:99-102``) -> 4 x [depthwise 3x3 + BN + ReLU,
pointwise 1x1 + BN + ReLU] -> ``AdaptiveAvgPool2d(1)`` -> dropout -> ``Linear(64,2)``.
Layer arithmetic = ``torch.nn`` semantics (BatchNorm2d eps 1e-5, momentum 0.1,
unbiased running var).  Module/parameter names are shared with the HIP-backed module
so ``state_dict``s interchange.

Dropout uses the counter-based Philox stream (oracle/philox.py) so the HIP kernel can
reproduce the keep-mask bit-exactly: keep(b,c) = u32 draw >= floor(p*2**32),
draw = philox(ctr=(step_lo, step_hi, b, TAG_DROPOUT<<24 | c//4))[c%4], scale 1/(1-p).
"""
from collections import OrderedDict
import numpy as np
import torch
import torch.nn as nn

from .philox import philox4x32_10, make_ctr, make_key, prob_threshold, TAG_DROPOUT


def dropout_keep_mask(B, C, p, seed, step, sample_offset=0):
    """(B,C) bool keep-mask."""
    if p <= 0.0:
        return np.ones((B, C), dtype=bool)
    b = (np.arange(B, dtype=np.uint64) + np.uint64(sample_offset))[:, None]
    q = np.arange((C + 3) // 4, dtype=np.uint64)[None, :]
    r = philox4x32_10(make_ctr(step, b, TAG_DROPOUT, q), make_key(seed)).astype(np.uint64)
    r = r.reshape(B, -1)[:, :C]
    return r >= np.uint64(prob_threshold(p))


class DSBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.dw = nn.Conv2d(ch, ch, 3, padding=1, groups=ch, bias=False)
        self.dw_bn = nn.BatchNorm2d(ch)
        self.pw = nn.Conv2d(ch, ch, 1, bias=False)
        self.pw_bn = nn.BatchNorm2d(ch)

    def forward(self, x):
        x = torch.relu(self.dw_bn(self.dw(x)))
        return torch.relu(self.pw_bn(self.pw(x)))


class CNNSmallOracle(nn.Module):
    def __init__(self, num_classes=2, dropout=0.3, input_channels=1, channels=64, n_blocks=4,
                 dropout_seed=0):
        super().__init__()
        self.stem = nn.Sequential(OrderedDict(
            conv=nn.Conv2d(input_channels, channels, 3, stride=2, padding=1, bias=False),
            bn=nn.BatchNorm2d(channels)))
        self.blocks = nn.ModuleList([DSBlock(channels) for _ in range(n_blocks)])
        self.classifier = nn.Linear(channels, num_classes)
        self.p = float(dropout)
        self.dropout_seed = int(dropout_seed)
        self.dropout_step = 0          # advanced once per training forward

    def features(self, x):
        x = torch.relu(self.stem(x))
        for blk in self.blocks:
            x = blk(x)
        return x.mean(dim=(2, 3))

    def forward(self, x):
        pooled = self.features(x)
        if self.training and self.p > 0.0:
            keep = dropout_keep_mask(pooled.shape[0], pooled.shape[1], self.p,
                                     self.dropout_seed, self.dropout_step)
            self.dropout_step += 1
            scale = np.float32(1.0 / (1.0 - float(np.float32(self.p))))
            pooled = pooled * torch.from_numpy(keep.astype(np.float32) * scale).to(pooled)
        return self.classifier(pooled)
