"""``SpecAugment`` with the constructor the reference's tests pin
(``tests/test_training_pipeline.py:252-257``: ``SpecAugment(freq_mask_param=15, time_mask_param=35,
n_freq_masks=2, n_time_masks=2)``, ``__call__((1,F,T)) -> same shape``) and the application probabilities of
``src/config/defaults.py:90-91``.  The mask law is integer-defined from a Philox4x32-10 counter stream
(DESIGN.md "SpecAugment spec") and runs in ``ww_specaug_apply`` (or fused into ``ww_logmel_fwd``).
Waveform-level augmentation (RIR / background mix, time-stretch, pitch) is outside this round's scope."""
import torch

from .. import _native as nat


class SpecAugment:
    def __init__(self, freq_mask_param: int = 15, time_mask_param: int = 35, n_freq_masks: int = 2,
                 n_time_masks: int = 2, freq_mask_prob: float = 1.0, time_mask_prob: float = 1.0, seed: int = 0):
        if min(freq_mask_param, time_mask_param, n_freq_masks, n_time_masks) < 0:
            raise ValueError("SpecAugment parameters must be non-negative")
        if n_freq_masks + n_time_masks > 16:
            raise ValueError("n_freq_masks + n_time_masks must be <= 16")
        self.freq_mask_param, self.time_mask_param = freq_mask_param, time_mask_param
        self.n_freq_masks, self.n_time_masks = n_freq_masks, n_time_masks
        self.freq_mask_prob, self.time_mask_prob = freq_mask_prob, time_mask_prob
        self.seed = seed
        self.step = 0                 # counter of the RNG stream; advanced once per call
        self.last_indices = None      # device int32 (B, n_f+n_t, 2) of the last call when return_indices

    def native_cfg(self):
        return nat.make_specaug_cfg(self.freq_mask_param, self.time_mask_param, self.n_freq_masks, self.n_time_masks,
                                    self.freq_mask_prob, self.time_mask_prob)

    def __call__(self, spec: torch.Tensor, step: int = None, sample_offset: int = 0, return_indices: bool = False):
        """spec (F,T)-batched as (B,F,T) / (B,1,F,T), float32 on an MI355X; returns a masked copy."""
        if spec.dim() not in (3, 4):
            raise ValueError(f"spectrogram must be (B,F,T) or (B,1,F,T), got {tuple(spec.shape)}")
        out = spec.float().clone().contiguous()
        st = self.step if step is None else step
        self.last_indices = nat.specaug_apply_(out, self.native_cfg(), seed=self.seed, step=st,
                                               sample_offset=sample_offset, want_idx=return_indices)
        if step is None:
            self.step += 1
        return out
