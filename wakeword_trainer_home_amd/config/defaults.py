"""Configuration tree read by the Trainer / factories.

Field names and default values follow the reference's dataclasses
(``src/config/defaults.py:12-147``) so configs interchange; only the fields the hot path
reads are consumed here (``trainer.py:79-83,92,96,107,112``, ``optimizer_factory.py:351-371``).
Additions (backward compatible, defaults keep reference behaviour): SpecAugment mask
geometry and the RNG seed in ``AugmentationConfig`` (the reference keeps them as
``SpecAugment`` ctor arguments, ``tests/test_training_pipeline.py:252-257``), and
``TrainingConfig.data_parallel``.
"""
from dataclasses import dataclass, field, asdict, fields
from pathlib import Path
from typing import Any, Dict, List

import yaml


class _Section:
    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @classmethod
    def from_mapping(cls, m):
        known = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in (m or {}).items() if k in known})


@dataclass
class DataConfig(_Section):
    sample_rate: int = 16000
    audio_duration: float = 2.5
    n_mfcc: int = 40
    n_fft: int = 1024
    hop_length: int = 160
    n_mels: int = 128
    feature_type: str = "mel"          # mel | mfcc
    normalize_audio: bool = True


@dataclass
class TrainingConfig(_Section):
    batch_size: int = 128
    epochs: int = 30
    learning_rate: float = 0.001
    early_stopping_patience: int = 10
    num_workers: int = 16
    pin_memory: bool = True
    persistent_workers: bool = True
    checkpoint_frequency: str = "best_only"   # best_only | every_epoch | every_5_epochs | every_10_epochs
    save_best_only: bool = True
    data_parallel: bool = True                # all-reduce gradients when torch.distributed is initialised
    deferred_metrics: bool = True             # HIP model: read a step's stats while the next step runs
    dp_overlap: Any = "auto"                  # data parallel: reduce the late layers' gradients under the early layers' backward
                                              # (True / False / "auto" = buckets >= 4 MB of bucketed autograd models, eager steps)
    hip_graph: bool = False                   # HIP model: capture the training step once per batch shape, replay it
    hip_graph_auto: bool = True               # ... and do so unasked for models that declare prefers_hip_graph (mobilenetv3: its
                                              # ~340 short launches per step are host-bound when issued eagerly, 5.2 vs 2.8 ms)


@dataclass
class ModelConfig(_Section):
    architecture: str = "resnet18"
    num_classes: int = 2
    pretrained: bool = True
    dropout: float = 0.3
    hidden_size: int = 128
    num_layers: int = 2
    bidirectional: bool = True


@dataclass
class AugmentationConfig(_Section):
    time_stretch_min: float = 0.80
    time_stretch_max: float = 1.20
    pitch_shift_min: int = -2
    pitch_shift_max: int = 2
    background_noise_prob: float = 0.5
    noise_snr_min: float = 5.0
    noise_snr_max: float = 20.0
    rir_prob: float = 0.25
    freq_mask_prob: float = 0.5
    time_mask_prob: float = 0.5
    # SpecAugment geometry (reference ctor values) + counter-RNG seed
    freq_mask_param: int = 15
    time_mask_param: int = 35
    n_freq_masks: int = 2
    n_time_masks: int = 2
    seed: int = 2024


@dataclass
class OptimizerConfig(_Section):
    optimizer: str = "adamw"
    weight_decay: float = 1e-4
    momentum: float = 0.9
    betas: List[float] = field(default_factory=lambda: [0.9, 0.999])
    scheduler: str = "cosine"
    warmup_epochs: int = 3
    min_lr: float = 3e-4
    step_size: int = 10
    gamma: float = 0.1
    patience: int = 15
    factor: float = 0.5
    gradient_clip: float = 1.0
    mixed_precision: bool = False
    amp_dtype: str = "bf16"          # HIP models: storage type behind mixed_precision -- "bf16" (no loss scaling needed) or
                                     # "fp16" (the reference's own AMP type, with the device-side GradScaler)


@dataclass
class LossConfig(_Section):
    loss_function: str = "cross_entropy"
    label_smoothing: float = 0.05
    focal_alpha: float = 0.25
    focal_gamma: float = 2.0
    class_weights: str = "balanced"
    hard_negative_weight: float = 2.5
    sampler_strategy: str = "weighted"


_SECTIONS = (("data", DataConfig), ("training", TrainingConfig), ("model", ModelConfig),
             ("augmentation", AugmentationConfig), ("optimizer", OptimizerConfig), ("loss", LossConfig))


@dataclass
class WakewordConfig:
    data: DataConfig = field(default_factory=DataConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    augmentation: AugmentationConfig = field(default_factory=AugmentationConfig)
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)
    loss: LossConfig = field(default_factory=LossConfig)
    config_name: str = "default"
    description: str = "Default wakeword training configuration"

    def to_dict(self) -> Dict[str, Any]:
        d = {"config_name": self.config_name, "description": self.description}
        for name, _ in _SECTIONS:
            d[name] = getattr(self, name).to_dict()
        return d

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "WakewordConfig":
        kw = {name: sec.from_mapping(d.get(name)) for name, sec in _SECTIONS}
        return cls(config_name=d.get("config_name", "default"), description=d.get("description", ""), **kw)

    def save(self, path):
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(yaml.safe_dump(self.to_dict(), default_flow_style=False, sort_keys=False))

    @classmethod
    def load(cls, path) -> "WakewordConfig":
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(f"Configuration file not found: {path}")
        return cls.from_dict(yaml.safe_load(path.read_text()))


def get_default_config() -> WakewordConfig:
    return WakewordConfig(config_name="default", description="Default balanced configuration for general use")
