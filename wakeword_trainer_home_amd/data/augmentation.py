"""``SpecAugment`` with the constructor the reference's tests pin
(``tests/test_training_pipeline.py:252-257``: ``SpecAugment(freq_mask_param=15, time_mask_param=35,
n_freq_masks=2, n_time_masks=2)``, ``__call__((1,F,T)) -> same shape``) and the application probabilities of
``src/config/defaults.py:90-91``.  The mask law is integer-defined from a Philox4x32-10 counter stream
(DESIGN.md "SpecAugment spec") and runs in ``ww_specaug_apply`` (or fused into ``ww_logmel_fwd``).
``AudioAugmentation`` keeps the constructor of ``tests/test_training_pipeline.py:230-236`` and runs the RIR convolution +
background-noise mix of BASELINE config 4 on the GPU (``ww_audio_augment``; law in DESIGN.md "Audio augmentation spec").
Time-stretch / pitch-shift are outside the north_star list: the ranges are accepted and stored, nothing is resampled."""
import torch

from .. import _native as nat


class AudioAugmentation:
    """``AudioAugmentation(sample_rate, device, time_stretch_range, pitch_shift_range, background_noise_prob, ...)``;
    ``__call__((B,N) waveform) -> (B,N)`` (shape/finite contract of tests/test_training_pipeline.py:242-243).
    ``rirs`` (R,L<=8192) and ``noises`` (K,Nn>=N) are float32 banks kept resident in HBM (the reference reads RIR / noise
    files per clip on CPU workers); without a bank the corresponding effect is off."""

    def __init__(self, sample_rate: int = 16000, device="cuda", time_stretch_range=(0.8, 1.2),
                 pitch_shift_range=(-2, 2), background_noise_prob: float = 0.5, noise_snr_range=(5.0, 20.0),
                 rir_prob: float = 0.25, rirs=None, noises=None, seed: int = 0, fft_min_taps: int = 129):
        for name, p in (("background_noise_prob", background_noise_prob), ("rir_prob", rir_prob)):
            if not 0.0 <= p <= 1.0:
                raise ValueError(f"{name} must be in [0, 1], got {p}")
        if noise_snr_range[1] < noise_snr_range[0]:
            raise ValueError("noise_snr_range must be (min, max) with max >= min")
        self.sample_rate = sample_rate
        self.device = torch.device(device)
        self.time_stretch_range, self.pitch_shift_range = tuple(time_stretch_range), tuple(pitch_shift_range)
        self.background_noise_prob, self.rir_prob = float(background_noise_prob), float(rir_prob)
        self.noise_snr_range = (float(noise_snr_range[0]), float(noise_snr_range[1]))
        self.seed = seed
        self.step = 0
        self.rirs = self._bank(rirs, "rirs")
        self.noises = self._bank(noises, "noises")
        # RIRs of fft_min_taps or more go through the overlap-save FFT form; their spectra are computed once, here
        self.rir_spectra = (nat.audio_rir_spectra(self.rirs)
                            if self.rirs is not None and self.rirs.shape[1] >= fft_min_taps else None)
        self.last_choices = None      # device int32 (B,4): rir|-1, noise|-1, offset, float bits of snr_db

    def _bank(self, bank, name):
        if bank is None:
            return None
        bank = torch.as_tensor(bank, dtype=torch.float32)
        if bank.dim() != 2 or bank.shape[0] < 1:
            raise ValueError(f"{name} must be a (count, length) array")
        return bank.to(self.device).contiguous()

    def __call__(self, audio: torch.Tensor, step: int = None, sample_offset: int = 0, return_choices: bool = False):
        if audio.dim() != 2:
            raise ValueError(f"waveform batch must be (B,N), got {tuple(audio.shape)}")
        wave = audio.to(self.device, non_blocking=True)
        wave = (wave.float() / 32768.0) if wave.dtype == torch.int16 else wave.float()
        st = self.step if step is None else step
        res = nat.audio_augment(wave.contiguous(), self.rirs, self.noises, self.rir_prob, self.background_noise_prob,
                                self.noise_snr_range[0], self.noise_snr_range[1], seed=self.seed, step=st,
                                sample_offset=sample_offset, want_choices=return_choices, rir_spectra=self.rir_spectra)
        if step is None:
            self.step += 1
        if return_choices:
            res, self.last_choices = res
        return res


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
