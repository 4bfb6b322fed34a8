"""Log-mel / MFCC oracle (float64 numpy + an fp32 torch.stft variant).

The reference's ``src/data/feature_extraction.py`` is ABSENT from the snapshot
(SURVEY.md F1); its API is known only from call sites
(``src/evaluation/evaluator.py:86-94,122-128``, ``src/evaluation/inference.py:94-102,194-200``)
and its output shape from ``src/export/onnx_exporter.py:316-320``.  The arithmetic
below is therefore the BUILD'S OWN SPEC (DESIGN.md "Feature spec"):
**parity unpinned** with respect to the reference.

Spec: periodic Hann(n_fft); center=True with reflect padding n_fft//2;
T = 1 + N // hop frames; rFFT(n_fft); power |X|^2; HTK mel triangular filterbank
(f_min=0, f_max=sr/2, norm=None), as published for torchaudio 2.1
``functional.melscale_fbanks``; log(mel + 1e-6) natural log.  MFCC = orthonormal
DCT-II of the log-mel over the mel axis, first n_mfcc coefficients.
"""
import numpy as np

LOG_EPS = 1e-6


def hann_periodic(n_fft: int) -> np.ndarray:
    k = np.arange(n_fft, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_fft)


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int,
                   f_min: float = 0.0, f_max: float = None) -> np.ndarray:
    """(n_freqs, n_mels) float64 triangular HTK filterbank, norm=None."""
    if f_max is None:
        f_max = sample_rate / 2.0
    all_freqs = np.linspace(0.0, sample_rate / 2.0, n_freqs)
    m_pts = np.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = mel_to_hz_htk(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]          # (n_freqs, n_mels+2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def dct_matrix(n_mfcc: int, n_mels: int) -> np.ndarray:
    """(n_mfcc, n_mels) orthonormal DCT-II."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    d = np.cos(np.pi / n_mels * (n + 0.5) * k)
    d[0] *= 1.0 / np.sqrt(2.0)
    return d * np.sqrt(2.0 / n_mels)


def frame_signal(x: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """x (B,N) -> (B,T,n_fft) with center=True reflect padding."""
    pad = n_fft // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    T = 1 + x.shape[1] // hop
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    return xp[:, idx]


def power_spectrogram(x, n_fft=1024, hop=160):
    x = np.asarray(x, dtype=np.float64)
    fr = frame_signal(x, n_fft, hop) * hann_periodic(n_fft)
    spec = np.fft.rfft(fr, n=n_fft, axis=-1)              # (B,T,F)
    return (spec.real ** 2 + spec.imag ** 2)


def logmel(x, sample_rate=16000, n_fft=1024, hop=160, n_mels=40,
           f_min=0.0, f_max=None, log_eps=LOG_EPS):
    """x (B,N) float -> (B,1,n_mels,T) float64."""
    p = power_spectrogram(x, n_fft, hop)                  # (B,T,F)
    fb = mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate, f_min, f_max)
    mel = p @ fb                                          # (B,T,M)
    return np.log(mel + log_eps).transpose(0, 2, 1)[:, None]


def mfcc(x, sample_rate=16000, n_fft=1024, hop=160, n_mels=40, n_mfcc=40,
         f_min=0.0, f_max=None, log_eps=LOG_EPS):
    lm = logmel(x, sample_rate, n_fft, hop, n_mels, f_min, f_max, log_eps)[:, 0]  # (B,M,T)
    d = dct_matrix(n_mfcc, n_mels)
    return np.einsum("cm,bmt->bct", d, lm)[:, None]


def logmel_torch(x, sample_rate=16000, n_fft=1024, hop=160, n_mels=40,
                 f_min=0.0, f_max=None, log_eps=LOG_EPS, n_mfcc=None):
    """fp32 torch.stft formulation of the same spec (the 'reference-style PyTorch
    CPU path' used for the cpu_baseline timing and as an independent cross-check)."""
    import torch
    x = torch.as_tensor(x, dtype=torch.float32)
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=win,
                      center=True, pad_mode="reflect", return_complex=True)   # (B,F,T)
    p = spec.real ** 2 + spec.imag ** 2
    fb = torch.from_numpy(mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate, f_min, f_max)).float()
    mel = torch.matmul(fb.t(), p)                         # (B,M,T)
    out = torch.log(mel + log_eps)
    if n_mfcc is not None:
        d = torch.from_numpy(dct_matrix(n_mfcc, n_mels)).float()
        out = torch.matmul(d, out)
    return out[:, None]
