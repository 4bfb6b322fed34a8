"""Metrics oracle: confusion-counter -> rates, restating
``src/training/metrics.py:112-153`` (``MetricsCalculator.calculate``).
PINNED: tests/golden/g3_metrics.json."""


def rates_from_counters(tp, tn, fp, fn):
    total = tp + tn + fp + fn
    acc = (tp + tn) / total if total > 0 else 0.0
    prec = tp / (tp + fp) if (tp + fp) > 0 else 0.0
    rec = tp / (tp + fn) if (tp + fn) > 0 else 0.0
    f1 = 2 * (prec * rec) / (prec + rec) if (prec + rec) > 0 else 0.0
    fpr = fp / (fp + tn) if (fp + tn) > 0 else 0.0
    fnr = fn / (fn + tp) if (fn + tp) > 0 else 0.0
    return dict(accuracy=acc, precision=prec, recall=rec, f1_score=f1, fpr=fpr, fnr=fnr,
                true_positives=tp, true_negatives=tn, false_positives=fp, false_negatives=fn,
                total_samples=total, positive_samples=tp + fn, negative_samples=tn + fp)
