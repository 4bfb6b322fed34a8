"""Presets the benchmark configurations name (BASELINE.json) plus the reference's own
names (``src/config/presets.py:336-343``).  Only knobs the training hot path reads differ
from the defaults; values follow the reference presets (small ``:35-92``, large ``:94-151``,
fast ``:154-211``)."""
from .defaults import WakewordConfig


def _default():
    return WakewordConfig(config_name="default", description="Default balanced configuration for general use")


def _small():
    c = WakewordConfig(config_name="small_dataset", description="Small datasets (<10k samples), aggressive augmentation")
    c.data.audio_duration = 1.5
    c.training.batch_size, c.training.epochs, c.training.learning_rate = 16, 100, 0.0005
    c.training.early_stopping_patience, c.training.num_workers = 15, 4
    c.model.architecture, c.model.pretrained, c.model.dropout = "mobilenetv3", True, 0.5
    c.augmentation.freq_mask_prob = c.augmentation.time_mask_prob = 0.7
    c.augmentation.background_noise_prob, c.augmentation.rir_prob = 0.7, 0.5
    c.optimizer.weight_decay = 1e-3
    c.loss.loss_function, c.loss.focal_alpha, c.loss.focal_gamma = "focal_loss", 0.25, 2.0
    return c


def _large():
    c = WakewordConfig(config_name="large_dataset", description="Large datasets (>100k samples), faster training")
    c.training.batch_size, c.training.epochs, c.training.learning_rate = 128, 30, 0.002
    c.training.early_stopping_patience = 8
    c.model.dropout = 0.2
    c.augmentation.background_noise_prob, c.augmentation.noise_snr_min = 0.4, 10.0
    c.augmentation.freq_mask_prob = c.augmentation.time_mask_prob = 0.2
    c.optimizer.mixed_precision = True
    c.loss.hard_negative_weight = 2.0
    return c


def _fast():
    c = WakewordConfig(config_name="fast_training", description="Quick iteration / prototyping")
    c.data.audio_duration, c.data.n_mels = 1.5, 64
    c.training.batch_size, c.training.epochs, c.training.learning_rate = 64, 20, 0.002
    c.training.early_stopping_patience = 5
    c.model.architecture, c.model.pretrained, c.model.dropout = "mobilenetv3", True, 0.2
    c.optimizer.scheduler, c.optimizer.warmup_epochs, c.optimizer.mixed_precision = "step", 0, True
    return c


def _cnn_small_bench():
    """BASELINE.json config 2: cnn_small + log-mel(40), 16 kHz x 1.5 s clips."""
    c = WakewordConfig(config_name="cnn_small_logmel40", description="cnn_small on 40-bin log-mel, 1.5 s clips")
    c.data.audio_duration, c.data.n_mels = 1.5, 40
    c.training.batch_size = 512
    c.model.architecture, c.model.pretrained = "cnn_small", False
    return c


PRESETS = {"default": _default, "small_dataset": _small, "large_dataset": _large, "fast_training": _fast,
           "cnn_small_logmel40": _cnn_small_bench}


def get_preset(preset_name: str) -> WakewordConfig:
    key = preset_name.lower().replace(" ", "_")
    if key not in PRESETS:
        raise ValueError(f"Unknown preset: {preset_name}. Available presets: {', '.join(PRESETS)}")
    return PRESETS[key]()


def list_presets():
    return list(PRESETS)
