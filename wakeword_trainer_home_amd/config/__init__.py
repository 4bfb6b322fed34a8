from .defaults import (DataConfig, TrainingConfig, ModelConfig, AugmentationConfig, OptimizerConfig, LossConfig,
                       WakewordConfig, get_default_config)
from .presets import get_preset, list_presets
from .cuda_utils import enforce_cuda

__all__ = ["DataConfig", "TrainingConfig", "ModelConfig", "AugmentationConfig", "OptimizerConfig", "LossConfig",
           "WakewordConfig", "get_default_config", "get_preset", "list_presets", "enforce_cuda"]
