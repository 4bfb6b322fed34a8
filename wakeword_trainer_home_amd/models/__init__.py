from .architectures import create_model, CNNSmallWakeword
from .losses import create_loss_function, LabelSmoothingCrossEntropy, CrossEntropyLoss, FocalLoss

__all__ = ["create_model", "CNNSmallWakeword", "create_loss_function", "LabelSmoothingCrossEntropy",
           "CrossEntropyLoss", "FocalLoss"]
