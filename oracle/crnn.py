"""Oracle of the CRNN (conv front-end + GRU; the reference has none -- SURVEY.md F4 -- so the topology is the build's,
DESIGN.md): ``oracle.cnn_small``'s conv stack (torch.nn Conv2d / BatchNorm2d / ReLU) -> mean over the frequency axis ->
``oracle.gru.GRUWakewordOracle``.  Test infrastructure only."""
import torch
import torch.nn as nn

from .cnn_small import CNNSmallOracle
from .gru import GRUWakewordOracle


class CRNNOracle(nn.Module):
    def __init__(self, hidden_size=128, num_layers=2, num_classes=2, bidirectional=True, dropout=0.3, seed=0,
                 dtype=torch.float64):
        super().__init__()
        self.front = CNNSmallOracle(dropout=0.0).to(dtype)
        self.rnn = GRUWakewordOracle(64, hidden_size, num_layers, num_classes, bidirectional, dropout, seed, dtype)
        self.dtype = dtype

    def load_device_state_dict(self, sd):
        """state_dict of wakeword_trainer_home_amd.models.recurrent.CRNNWakeword (front.* / rnn.*)."""
        front = {k[len("front."):]: v for k, v in sd.items() if k.startswith("front.")}
        cur = self.front.state_dict()
        cur.update({k: v.to(cur[k].dtype) for k, v in front.items()})
        self.front.load_state_dict(cur)
        self.rnn.load_reference_state_dict({k[len("rnn."):]: v for k, v in sd.items() if k.startswith("rnn.")})

    def forward(self, x, step=0, sample_offset=0, training=True):
        f = self.front
        h = torch.relu(f.stem.bn(f.stem.conv(x.to(self.dtype))))
        for blk in f.blocks:
            h = blk(h)
        seq = h.mean(dim=2).transpose(1, 2)            # (B,64,H,W) -> (B,W,64)
        return self.rnn(seq, step=step, sample_offset=sample_offset, training=training)
