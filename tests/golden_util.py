"""Helpers shared by CPU and GPU parity tests for reading the G2 trace fixture."""
import json

import numpy as np
import torch


def make_inputs(seed, n, F=40, T=151):
    """Must mirror tests/golden/make_golden.py:make_inputs (seeded CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, F, T, generator=g) * 2.0 - 4.0
    y = (torch.rand(n, generator=g) < 0.3).long()
    return x, y


def load_trace(golden_dir, tag):
    meta = json.loads((golden_dir / "g2_meta.json").read_text())[tag]
    z = np.load(golden_dir / "g2_trace.npz")
    tr = dict(step_loss=z[f"{tag}/step_loss"], step_acc=z[f"{tag}/step_acc"],
              grad_norm=z[f"{tag}/grad_norm"], init={}, final={})
    for k in z.files:
        if k.startswith(f"{tag}/init/"):
            tr["init"][k[len(tag) + 6:]] = z[k]
        elif k.startswith(f"{tag}/final/"):
            tr["final"][k[len(tag) + 7:]] = z[k]
    return meta, tr


def build_optimizer(model, cfg):
    o, t = cfg["optimizer"], cfg["training"]
    if o["optimizer"] == "adamw":
        return torch.optim.AdamW(model.parameters(), lr=t["learning_rate"], betas=tuple(o["betas"]),
                                 weight_decay=o["weight_decay"])
    if o["optimizer"] == "adam":
        return torch.optim.Adam(model.parameters(), lr=t["learning_rate"], betas=tuple(o["betas"]),
                                weight_decay=o["weight_decay"])
    return torch.optim.SGD(model.parameters(), lr=t["learning_rate"], momentum=o["momentum"],
                           weight_decay=o["weight_decay"], nesterov=True)
