"""Loss factory of the hot path: same names, constructor arguments and input validation as
the reference (``src/models/losses.py:14-270``); the arithmetic (loss, gradient, batch
counters) is one HIP kernel, ``ww_ce2_loss_fwd_bwd``.  Scope: 2 classes, mean reduction, no
per-class weights -- what ``Trainer`` actually uses (``src/training/trainer.py:78-86`` passes
``class_weights=None``)."""
from typing import Optional

import torch
import torch.nn as nn

from .. import _native as nat


class _CE2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, mod):
        loss, dl, stats = nat.ce2_loss_fwd_bwd(logits.contiguous(), targets.contiguous(), mod._kind, mod._eps,
                                               mod._alpha, mod._gamma, stats=mod._stats_for(logits.device),
                                               found_inf_out=mod.found_inf_out, loss_scale=mod.loss_scale,
                                               loss_scale_slot=mod.loss_scale_slot)
        mod.last_stats = stats
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        (dl,) = ctx.saved_tensors
        return dl * gout, None, None


class _NativeLoss(nn.Module):
    _kind = nat.LOSS_CE
    _eps, _alpha, _gamma = 0.0, 0.25, 2.0

    def __init__(self, weight: Optional[torch.Tensor], reduction: str):
        super().__init__()
        if weight is not None:
            raise ValueError("per-class weights are not implemented in the HIP loss kernel "
                             "(the reference Trainer never passes them: trainer.py:84)")
        if reduction != "mean":
            raise ValueError(f"the HIP loss kernel implements reduction='mean', got {reduction!r}")
        self.weight, self.reduction = weight, reduction
        self.validate_targets = True      # reference behaviour: raise at once (costs a host sync)
        self.last_stats = None            # device uint8[48] = ww_step_stats of the last call
        self.found_inf_out = None         # float32[1] device tensor that also receives found_inf (data-parallel bucket slot)
        self.loss_scale = None            # device ww_loss_scale (fp16 storage): dL/dlogits leaves the kernel times scale[slot]
        self.loss_scale_slot = 0
        self._stats = {}

    def _stats_for(self, dev):
        if dev not in self._stats:
            self._stats[dev] = torch.zeros(nat.STEP_STATS_BYTES, dtype=torch.uint8, device=dev)
        return self._stats[dev]

    def native_fwd_bwd(self, logits: torch.Tensor, target: torch.Tensor, found_inf_out=None):
        """Loss kernel without autograd: -> (device ww_step_stats, dL/dlogits).  Target range errors are reported through
        ``stats.bad_target`` (the caller reads the stats once per step)."""
        if logits.dim() != 2 or logits.size(1) != 2 or target.dim() != 1 or logits.size(0) != target.size(0):
            raise ValueError(f"expected logits (B,2) and targets (B,), got {tuple(logits.shape)} and {tuple(target.shape)}")
        _, dl, stats = nat.ce2_loss_fwd_bwd(logits.contiguous(), target.long().contiguous(), self._kind, self._eps,
                                            self._alpha, self._gamma, stats=self._stats_for(logits.device),
                                            found_inf_out=found_inf_out, loss_scale=self.loss_scale,
                                            loss_scale_slot=self.loss_scale_slot)
        self.last_stats = stats
        return stats, dl

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if pred.dim() != 2:
            raise ValueError(f"Predictions must be 2D (batch, num_classes), got shape {pred.shape}")
        if target.dim() != 1:
            raise ValueError(f"Targets must be 1D (batch,), got shape {target.shape}")
        if pred.size(0) != target.size(0):
            raise ValueError(f"Batch size mismatch: pred={pred.size(0)}, target={target.size(0)}")
        if pred.size(1) != 2:
            raise ValueError(f"the HIP loss kernel implements num_classes == 2, got {pred.size(1)}")
        if not pred.is_cuda:
            raise nat.NativeError("the loss runs on a hand-written HIP kernel only; tensors must be on an MI355X "
                                  "('cuda') device -- there is no CPU fallback")
        loss = _CE2Fn.apply(pred.float(), target.long(), self)
        if self.validate_targets:
            st = nat.decode_stats(self.last_stats.cpu())
            if st["bad_target"]:
                raise ValueError("Target values must be in [0, 1]")
        return loss


class LabelSmoothingCrossEntropy(_NativeLoss):
    def __init__(self, smoothing: float = 0.1, weight: Optional[torch.Tensor] = None, reduction: str = "mean"):
        if not 0.0 <= smoothing <= 1.0:
            raise ValueError(f"Label smoothing must be in [0, 1], got {smoothing}")
        super().__init__(weight, reduction)
        self.smoothing, self.confidence = smoothing, 1.0 - smoothing
        self._eps = float(smoothing)


class CrossEntropyLoss(LabelSmoothingCrossEntropy):
    """label_smoothing == 0 branch (the reference returns nn.CrossEntropyLoss there, losses.py:256)."""

    def __init__(self, weight: Optional[torch.Tensor] = None, reduction: str = "mean"):
        super().__init__(0.0, weight, reduction)


class FocalLoss(_NativeLoss):
    _kind = nat.LOSS_FOCAL
    EPS = 1e-7

    def __init__(self, alpha: float = 0.25, gamma: float = 2.0, weight: Optional[torch.Tensor] = None,
                 reduction: str = "mean"):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Alpha must be in [0, 1], got {alpha}")
        if gamma < 0:
            raise ValueError(f"Gamma must be non-negative, got {gamma}")
        super().__init__(weight, reduction)
        self.alpha, self.gamma = alpha, gamma
        self._alpha, self._gamma = float(alpha), float(gamma)


def create_loss_function(loss_name: str, num_classes: int = 2, label_smoothing: float = 0.1,
                         focal_alpha: float = 0.25, focal_gamma: float = 2.0,
                         class_weights: Optional[torch.Tensor] = None, device: str = "cuda") -> nn.Module:
    name = loss_name.lower()
    if num_classes != 2 and name in ("cross_entropy", "focal_loss"):
        raise ValueError(f"the HIP loss kernels implement num_classes == 2, got {num_classes}")
    if name == "cross_entropy":
        if label_smoothing > 0:
            return LabelSmoothingCrossEntropy(smoothing=label_smoothing, weight=class_weights, reduction="mean")
        return CrossEntropyLoss(weight=class_weights, reduction="mean")
    if name == "focal_loss":
        return FocalLoss(alpha=focal_alpha, gamma=focal_gamma, weight=class_weights, reduction="mean")
    raise ValueError(f"Unknown loss function: {name}. Supported: cross_entropy, focal_loss")
