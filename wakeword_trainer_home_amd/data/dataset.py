"""Synthetic-clip loader for benchmarks and tests (BASELINE.md §4 / SURVEY.md §8d): waveforms
``x ~ N(0, 0.1^2)`` clipped to [-1, 1], labels Bernoulli(0.1).  It honours the reference's batch contract
``(inputs, targets, metadata)`` (``src/evaluation/evaluator.py:257-266``; ``src/training/trainer.py:154-157``).
File decoding / manifests of the reference's (absent) ``WakewordDataset`` are outside the hot path."""
import torch
from torch.utils.data import Dataset


def make_synthetic_batch(batch, n_samples=24000, seed=1234, device="cpu", pos_rate=0.1, dtype=torch.float32):
    g = torch.Generator(device=device).manual_seed(seed)
    x = (torch.randn(batch, n_samples, generator=g, device=device) * 0.1).clamp_(-1.0, 1.0)
    y = (torch.rand(batch, generator=g, device=device) < pos_rate).long()
    if dtype == torch.int16:
        x = (x * 32767.0).round().to(torch.int16)
    return x, y


class SyntheticClipDataset(Dataset):
    """Pre-generated clips held in one tensor (optionally already on the GPU)."""

    def __init__(self, n_clips, n_samples=24000, seed=1234, device="cpu", pos_rate=0.1):
        self.wave, self.label = make_synthetic_batch(n_clips, n_samples, seed, device, pos_rate)

    def __len__(self):
        return self.wave.shape[0]

    def __getitem__(self, i):
        return self.wave[i], int(self.label[i]), {"path": f"synthetic://{i}"}
