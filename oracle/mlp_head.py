"""Oracle of the dense classifier head: the ``classifier`` the reference puts on MobileNetV3,
``nn.Sequential(Linear(576,1024), Hardswish(), Dropout(p), Linear(1024,num_classes))`` (src/models/architectures.py:105-111),
in plain torch on the CPU.  torch's own Linear / Hardswish define the arithmetic; only the dropout mask is the build's
(Philox, oracle/cnn_small.py:dropout_keep_mask -- torch's RNG stream cannot be reproduced on the device), so the mask is
applied explicitly.  ``bf16=True`` restates the device's reduced-precision mode: GEMM operands rounded to bf16, products
and sums in fp32.  Test infrastructure only."""
import numpy as np
import torch
import torch.nn as nn

from .cnn_small import dropout_keep_mask


def _r(t, bf16):
    return t.bfloat16().float() if bf16 else t


class MLPHeadOracle(nn.Module):
    def __init__(self, in_features=576, hidden=1024, num_classes=2, dropout=0.3, seed=0, dtype=torch.float64):
        super().__init__()
        self.classifier = nn.Sequential(nn.Linear(in_features, hidden), nn.Hardswish(), nn.Dropout(dropout),
                                        nn.Linear(hidden, num_classes)).to(dtype)
        self.p, self.seed, self.dtype = float(np.float32(dropout)), seed, dtype

    def forward(self, x, step=0, sample_offset=0, training=True, bf16=False):
        l0, l3 = self.classifier[0], self.classifier[3]
        x = x.to(self.dtype)
        pre = _r(x, bf16) @ _r(l0.weight, bf16).t() + l0.bias
        h = torch.nn.functional.hardswish(pre)
        if training and self.p > 0:
            keep = torch.from_numpy(dropout_keep_mask(x.shape[0], pre.shape[1], self.p, self.seed, step, sample_offset))
            h = h * keep.to(self.dtype) * (1.0 / (1.0 - self.p))
        return _r(h, bf16) @ _r(l3.weight, bf16).t() + l3.bias
