from .trainer import Trainer, TrainingState
from .metrics import MetricResults, MetricsCalculator, MetricsTracker, MetricMonitor, calculate_class_weights
from .optimizer_factory import (create_optimizer, create_scheduler, create_optimizer_and_scheduler,
                                create_grad_scaler, clip_gradients, get_learning_rate, WarmupScheduler)

__all__ = ["Trainer", "TrainingState", "MetricResults", "MetricsCalculator", "MetricsTracker", "MetricMonitor",
           "calculate_class_weights", "create_optimizer", "create_scheduler", "create_optimizer_and_scheduler",
           "create_grad_scaler", "clip_gradients", "get_learning_rate", "WarmupScheduler"]
