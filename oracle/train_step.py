"""Inner-step oracle: the body of ``src/training/trainer.py:165-203`` restated on
plain PyTorch CPU (fp32, AMP off): zero_grad -> forward -> loss -> finite check ->
backward -> clip_grad_norm_ (``optimizer_factory.py:446-452``) -> optimizer.step ->
batch accuracy.  Used by tests (against the G2 trace captured from the real reference
``Trainer``) and by ``bench.py``'s ``cpu_baseline`` leg.
PINNED: tests/golden/g2_trace.npz."""
import math
import numpy as np
import torch

from . import features as feat
from .specaugment import specaug_indices, specaug_apply
from .losses import ce_label_smoothing, focal


class TorchLoss(torch.nn.Module):
    """torch version of oracle/losses.py (autograd supplies the gradient)."""

    def __init__(self, kind="cross_entropy", eps=0.05, alpha=0.25, gamma=2.0):
        super().__init__()
        self.kind, self.eps, self.alpha, self.gamma = kind, eps, alpha, gamma

    def forward(self, z, y):
        lp = torch.log_softmax(z, dim=-1)
        oh = torch.nn.functional.one_hot(y, z.shape[-1]).to(z.dtype)
        if self.kind == "cross_entropy":
            s = oh * (1.0 - self.eps) + (1.0 - oh) * (self.eps / (z.shape[-1] - 1))
            return -(s * lp).sum(-1).mean()
        pt = (lp.exp() * oh).sum(-1).clamp(1e-7, 1.0 - 1e-7)
        a_t = torch.where(y == 1, torch.tensor(self.alpha), torch.tensor(1.0 - self.alpha))
        return (a_t * (1 - pt) ** self.gamma * (-(lp * oh).sum(-1))).mean()


def frontend(wave, spec_cfg=None, seed=0, step=0, n_mels=40, n_fft=1024, hop=160, sr=16000):
    """(B,N) waveform -> (B,1,M,T) fp32 features (+ SpecAugment) on CPU via torch.stft."""
    x = feat.logmel_torch(wave, sr, n_fft, hop, n_mels)
    idx = None
    if spec_cfg is not None:
        B, _, F, T = x.shape
        idx = specaug_indices(B, F, T, seed=seed, step=step, **spec_cfg)
        x = torch.from_numpy(specaug_apply(x.numpy(), idx, spec_cfg["n_freq_masks"]))
    return x, idx


def train_step(model, criterion, optimizer, inputs, targets, gradient_clip=1.0):
    """One reference-style optimisation step.  Returns dict(loss, acc, grad_norm, skipped)."""
    optimizer.zero_grad(set_to_none=True)
    outputs = model(inputs)
    loss = criterion(outputs, targets)
    if not torch.isfinite(loss):
        return dict(loss=float(loss), acc=float("nan"), grad_norm=0.0, skipped=True)
    loss.backward()
    gn = 0.0
    if gradient_clip > 0:
        gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_clip))
    optimizer.step()
    with torch.no_grad():
        acc = float((outputs.argmax(1) == targets).float().mean())
    return dict(loss=float(loss.detach()), acc=acc, grad_norm=gn, skipped=False)
