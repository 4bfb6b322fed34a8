"""Binary-classification metrics with the reference's result types and semantics
(``src/training/metrics.py``): ``MetricResults`` fields/``to_dict`` (:16-63), rate formulas with the
same zero-denominator conventions (:112-153), epoch tracker API (:187-290), sliding-window monitor
(:322-376), class weights (:379-427).

Difference by design: the reference appends every batch's logits/targets to CPU lists (a D2H copy +
sync per step, :219-220) and recounts at epoch end.  Here the tracker accumulates the five confusion
counters the loss kernel already produced (``ww_step_stats``); because ``calculate`` is argmax-based for
2-D predictions (:105-107) the epoch ``MetricResults`` are identical.
"""
import logging
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)


@dataclass
class MetricResults:
    accuracy: float
    precision: float
    recall: float
    f1_score: float
    fpr: float
    fnr: float
    true_positives: int
    true_negatives: int
    false_positives: int
    false_negatives: int
    total_samples: int
    positive_samples: int
    negative_samples: int

    def __str__(self) -> str:
        return (f"Accuracy: {self.accuracy:.4f} | Precision: {self.precision:.4f} | Recall: {self.recall:.4f} | "
                f"F1: {self.f1_score:.4f} | FPR: {self.fpr:.4f} | FNR: {self.fnr:.4f}")

    def to_dict(self) -> Dict[str, float]:
        return {k: getattr(self, k) for k in self.__dataclass_fields__}

    @classmethod
    def from_counts(cls, tp: int, tn: int, fp: int, fn: int) -> "MetricResults":
        def ratio(a, b):
            return a / b if b > 0 else 0.0
        total = tp + tn + fp + fn
        precision, recall = ratio(tp, tp + fp), ratio(tp, tp + fn)
        f1 = ratio(2 * precision * recall, precision + recall)
        return cls(accuracy=ratio(tp + tn, total), precision=precision, recall=recall, f1_score=f1,
                   fpr=ratio(fp, fp + tn), fnr=ratio(fn, fn + tp), true_positives=tp, true_negatives=tn,
                   false_positives=fp, false_negatives=fn, total_samples=total, positive_samples=tp + fn,
                   negative_samples=tn + fp)

    @classmethod
    def empty(cls) -> "MetricResults":
        return cls.from_counts(0, 0, 0, 0)


def _counts(predictions: torch.Tensor, targets: torch.Tensor, threshold: float) -> Tuple[int, int, int, int]:
    pred = predictions.argmax(dim=1) if predictions.dim() == 2 else (predictions > threshold).long()
    pred, targets = pred.to(targets.device), targets
    packed = torch.stack([((pred == 1) & (targets == 1)).sum(), ((pred == 0) & (targets == 0)).sum(),
                          ((pred == 1) & (targets == 0)).sum(), ((pred == 0) & (targets == 1)).sum()])
    tp, tn, fp, fn = (int(v) for v in packed.tolist())      # one host read
    return tp, tn, fp, fn


class MetricsCalculator:
    def __init__(self, device: str = "cuda"):
        self.device = device

    def calculate(self, predictions: torch.Tensor, targets: torch.Tensor, threshold: float = 0.5) -> MetricResults:
        tp, tn, fp, fn = _counts(predictions, targets, threshold)
        res = MetricResults.from_counts(tp, tn, fp, fn)
        res.total_samples = len(targets)                      # reference: len(targets), labels outside {0,1} count here
        res.accuracy = (tp + tn) / res.total_samples if res.total_samples > 0 else 0.0
        return res

    def confusion_matrix(self, predictions: torch.Tensor, targets: torch.Tensor, num_classes: int = 2) -> torch.Tensor:
        pred = predictions.argmax(dim=1) if predictions.dim() == 2 else predictions.long()
        idx = targets.long() * num_classes + pred.long().to(targets.device)
        return torch.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


class MetricsTracker:
    def __init__(self, device: str = "cuda"):
        self.device = device
        self.calculator = MetricsCalculator(device=device)
        self.epoch_metrics: List[MetricResults] = []
        self.reset()

    def reset(self):
        self._c = [0, 0, 0, 0]
        self._seen = False

    def update_counts(self, tp: int, tn: int, fp: int, fn: int):
        """Fast path: counters already produced on the device (ww_step_stats)."""
        for i, v in enumerate((tp, tn, fp, fn)):
            self._c[i] += int(v)
        self._seen = True

    def update(self, predictions: torch.Tensor, targets: torch.Tensor):
        """Reference-compatible entry: reduce this batch to its four counters right away."""
        self.update_counts(*_counts(predictions.detach(), targets.detach(), 0.5))

    def compute(self, threshold: float = 0.5) -> MetricResults:
        if not self._seen:
            logger.warning("No predictions accumulated, returning zero metrics")
            return MetricResults.empty()
        return MetricResults.from_counts(*self._c)

    def save_epoch_metrics(self, metrics: MetricResults):
        self.epoch_metrics.append(metrics)

    def get_epoch_history(self) -> List[MetricResults]:
        return self.epoch_metrics

    def get_best_epoch(self, metric: str = "f1_score") -> Tuple[int, Optional[MetricResults]]:
        if not self.epoch_metrics:
            return 0, None
        vals = [getattr(m, metric) for m in self.epoch_metrics]
        best = int(np.argmin(vals)) if metric in ("fpr", "fnr") else int(np.argmax(vals))
        return best, self.epoch_metrics[best]

    def summary(self) -> str:
        if not self.epoch_metrics:
            return "No metrics recorded"
        lines = ["METRICS SUMMARY", "=" * 80, ""]
        lines += [f"Epoch {i + 1:3d}: {m}" for i, m in enumerate(self.epoch_metrics)]
        lines += ["", "=" * 80]
        for label, key in (("Best Accuracy", "accuracy"), ("Best F1 Score", "f1_score"), ("Best FPR (lowest)", "fpr")):
            i, m = self.get_best_epoch(key)
            lines.append(f"{label}: {getattr(m, key):.4f} (Epoch {i + 1})")
        return "\n".join(lines)


class MetricMonitor:
    def __init__(self, window_size: int = 100):
        self.window_size = window_size
        self.batch_losses: List[float] = []
        self.batch_accuracies: List[float] = []

    def update_batch(self, loss: float, accuracy: float):
        self.batch_losses = (self.batch_losses + [loss])[-self.window_size:]
        self.batch_accuracies = (self.batch_accuracies + [accuracy])[-self.window_size:]

    def get_running_averages(self) -> Dict[str, float]:
        if not self.batch_losses:
            return {"loss": 0.0, "accuracy": 0.0}
        return {"loss": float(np.mean(self.batch_losses)), "accuracy": float(np.mean(self.batch_accuracies))}

    def reset(self):
        self.batch_losses.clear()
        self.batch_accuracies.clear()


def calculate_class_weights(dataset_stats: Dict[str, int], method: str = "balanced", device: str = "cuda") -> torch.Tensor:
    pos, neg = dataset_stats.get("positive", 0), dataset_stats.get("negative", 0)
    if pos == 0 or neg == 0:
        logger.warning("Zero samples in one class, returning equal weights")
        return torch.ones(2, device=device)
    total = pos + neg
    if method == "balanced":
        w_pos, w_neg = total / (2 * pos), total / (2 * neg)
    elif method == "inverse":
        w_pos, w_neg = neg / pos, 1.0
    elif method == "sqrt_inverse":
        w_pos, w_neg = float(np.sqrt(neg / pos)), 1.0
    else:
        raise ValueError(f"Unknown weighting method: {method}")
    return torch.tensor([w_neg, w_pos], device=device)
