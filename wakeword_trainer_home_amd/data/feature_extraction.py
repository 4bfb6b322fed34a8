"""``FeatureExtractor`` with the call contract of the reference's (absent) ``src/data/feature_extraction.py``
as its callers use it (``src/evaluation/evaluator.py:86-94,122-128``, ``src/evaluation/inference.py:94-102,194-200``):
``FeatureExtractor(sample_rate, feature_type, n_mels, n_mfcc, n_fft, hop_length, device)`` and
``extractor(waveform_1d) -> (1, n_feat, T)`` with ``T = N // hop + 1`` (``src/export/onnx_exporter.py:316-320``).
Batched extension: a ``(B, N)`` tensor returns ``(B, 1, n_feat, T)``.  Arithmetic: DESIGN.md "Feature spec",
executed by the fused HIP kernel ``ww_logmel_fwd``; no torchaudio/librosa path exists here."""
import numpy as np
import torch

from .. import _native as nat


class FeatureExtractor:
    def __init__(self, sample_rate: int = 16000, feature_type: str = "mel", n_mels: int = 128, n_mfcc: int = 40,
                 n_fft: int = 1024, hop_length: int = 160, device: str = "cuda", f_min: float = 0.0,
                 f_max: float = 0.0, log_eps: float = 1e-6):
        if feature_type == "mel_spectrogram":        # legacy alias the reference callers still map (evaluator.py:82-83)
            feature_type = "mel"
        if feature_type not in ("mel", "mfcc"):
            raise ValueError(f"Unknown feature_type: {feature_type}. Supported: mel, mfcc")
        self.sample_rate, self.feature_type, self.n_mels, self.n_mfcc = sample_rate, feature_type, n_mels, n_mfcc
        self.n_fft, self.hop_length, self.device = n_fft, hop_length, device
        self._cfg = nat.make_feat_cfg(sample_rate=sample_rate, n_fft=n_fft, hop=hop_length, n_mels=n_mels,
                                      n_mfcc=n_mfcc if feature_type == "mfcc" else 0, f_min=f_min, f_max=f_max,
                                      log_eps=log_eps)

    @property
    def n_features(self) -> int:
        return self.n_mfcc if self.feature_type == "mfcc" else self.n_mels

    def num_frames(self, n_samples: int) -> int:
        return n_samples // self.hop_length + 1

    def __call__(self, waveform, specaug=None, seed: int = 0, step: int = 0, sample_offset: int = 0):
        if isinstance(waveform, np.ndarray):
            waveform = torch.from_numpy(waveform)
        single = waveform.dim() == 1
        if single:
            waveform = waveform.unsqueeze(0)
        if waveform.dim() != 2:
            raise ValueError(f"waveform must be (N,) or (B,N), got {tuple(waveform.shape)}")
        if waveform.dtype not in (torch.float32, torch.int16):
            waveform = waveform.float()
        wave = waveform.to(self.device).contiguous()
        sa = specaug.native_cfg() if specaug is not None else None
        out = nat.logmel_fwd(wave, self._cfg, sa, seed=seed, step=step, sample_offset=sample_offset)
        return out[0] if single else out
