"""Oracle of the recurrent model: the reference's ``GRUWakeword`` (src/models/architectures.py:198-267: nn.GRU(input, 128,
num_layers, batch_first=True, dropout, bidirectional) -> final hidden states of the last layer concatenated -> Dropout ->
Linear) in plain torch on the CPU.  The GRU arithmetic is torch.nn.GRU's own; torch's dropout RNG cannot be reproduced on
the device, so the stack is unrolled into single-layer nn.GRU modules with the build's Philox masks applied explicitly
between layers and in front of ``fc`` (mask law: ww_dropout_bt, include/wwhip.h).  ``load_reference_state_dict`` takes a
state_dict with the reference's keys (``gru.weight_ih_l0`` ... ``fc.1.bias``).  Test infrastructure only."""
import numpy as np
import torch
import torch.nn as nn

from .philox import philox4x32_10, make_key, prob_threshold

TAG_DROPOUT = 1


def dropout_bt_mask(B, T, C, p, seed=0, step=0, sample_offset=0, stream_id=0):
    """(B,T,C) bool keep-mask of ww_dropout_bt."""
    if p <= 0.0:
        return np.ones((B, T, C), dtype=bool)
    cq = (C + 3) // 4
    b = (np.arange(B, dtype=np.uint64) + np.uint64(sample_offset))[:, None, None]
    t = np.arange(T, dtype=np.uint64)[None, :, None]
    q = np.arange(cq, dtype=np.uint64)[None, None, :]
    field = (np.uint64(TAG_DROPOUT) << np.uint64(24)) | (np.uint64(stream_id) << np.uint64(20)) | (t << np.uint64(8)) | q
    ctr = np.empty((B, T, cq, 4), dtype=np.uint32)
    ctr[..., 0] = np.uint32(step & 0xFFFFFFFF)
    ctr[..., 1] = np.uint32((step >> 32) & 0xFFFFFFFF)
    ctr[..., 2] = np.broadcast_to(b, (B, T, cq)).astype(np.uint32)
    ctr[..., 3] = np.broadcast_to(field, (B, T, cq)).astype(np.uint32)
    r = philox4x32_10(ctr.reshape(-1, 4), make_key(seed)).astype(np.uint64).reshape(B, T, cq * 4)[:, :, :C]
    return r >= np.uint64(prob_threshold(p))


class GRUWakewordOracle(nn.Module):
    def __init__(self, input_size=40, hidden_size=128, num_layers=2, num_classes=2, bidirectional=True, dropout=0.3,
                 seed=0, dtype=torch.float64):
        super().__init__()
        nd = 2 if bidirectional else 1
        self.layers = nn.ModuleList([nn.GRU(input_size if k == 0 else nd * hidden_size, hidden_size, num_layers=1,
                                            batch_first=True, bidirectional=bidirectional) for k in range(num_layers)]).to(dtype)
        self.fc = nn.Linear(nd * hidden_size, num_classes).to(dtype)
        self.p = float(np.float32(dropout)) if num_layers > 1 else 0.0
        self.p_fc = float(np.float32(dropout))
        self.seed, self.dtype, self.nd, self.H = seed, dtype, nd, hidden_size

    def load_reference_state_dict(self, sd):
        for k, layer in enumerate(self.layers):
            for sfx in ("", "_reverse")[:self.nd]:
                for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    getattr(layer, f"{name}_l0{sfx}").data.copy_(sd[f"gru.{name}_l{k}{sfx}"].to(self.dtype))
        self.fc.weight.data.copy_(sd["fc.1.weight"].to(self.dtype))
        self.fc.bias.data.copy_(sd["fc.1.bias"].to(self.dtype))

    def forward(self, x, step=0, sample_offset=0, training=True):
        if x.dim() == 4:                                   # (B,1,F,T) features, as the Trainer hands them over
            x = x[:, 0].transpose(1, 2)
        x = x.to(self.dtype)
        B, T, _ = x.shape
        hn = None
        for k, layer in enumerate(self.layers):
            x, hn = layer(x)
            if training and self.p > 0 and k + 1 < len(self.layers):
                keep = torch.from_numpy(dropout_bt_mask(B, T, x.shape[2], self.p, self.seed, step, sample_offset, 1 + k))
                x = x * keep.to(self.dtype) * (1.0 / (1.0 - self.p))
        h = torch.cat([hn[0], hn[1]], dim=1) if self.nd == 2 else hn[0]
        if training and self.p_fc > 0:
            keep = torch.from_numpy(dropout_bt_mask(B, 1, h.shape[1], self.p_fc, self.seed, step, sample_offset, 15))[:, 0]
            h = h * keep.to(self.dtype) * (1.0 / (1.0 - self.p_fc))
        return self.fc(h)
