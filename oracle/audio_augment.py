"""Waveform augmentation oracle: RIR convolution + background-noise mix (BASELINE config 4).

The reference's ``AudioAugmentation`` source is ABSENT; only its constructor
(``tests/test_training_pipeline.py:230-236``), its shape/finite contract (``:242-243``), the config knobs
(``src/config/defaults.py:81-87``: background_noise_prob, noise_snr_min/max, rir_prob) and the UI guidance
(``src/ui/panel_docs.py:111-122``) survive.  The law below is the BUILD'S OWN SPEC (DESIGN.md "Audio augmentation
spec") -- **parity unpinned** w.r.t. the reference.  Time-stretch / pitch-shift are outside the north_star list.

Per clip b (global sample index g = sample_offset + b), two Philox4x32-10 draws with ctr = (step_lo, step_hi, g,
TAG_AUDIO<<24 | i), key = seed:
  i = 0 (RIR):    apply = r0 < floor(float32(rir_prob) * 2**32);  rir = r1 mod R
  i = 1 (noise):  apply = r0 < floor(float32(noise_prob) * 2**32); noise = r1 mod K;  offset = r2 mod (Nn - N + 1);
                  u = float32(r3 >> 8) * 2**-24;  snr_db = fma(u, snr_max - snr_min, snr_min)          (float32)
Signal law:
  y  = (x * h_rir)[0:N]  (causal linear convolution, zero history) if apply_rir else x
  y *= rms(x) / rms(y)   (RIR keeps the clip's loudness; skipped when rms(y) == 0)
  n  = noise[noise][offset : offset + N];  gain = rms(y) / (rms(n) * 10**(snr_db/20))   (0 if rms(n) == 0)
  out = clip(y + gain * n, -1, 1)         (no noise -> out = clip(y, -1, 1))
"""
import numpy as np

from .philox import philox4x32_10, make_ctr, make_key, prob_threshold

TAG_AUDIO = 2


def audio_choices(B, N, R, K, Nn, rir_prob, noise_prob, snr_min, snr_max, seed=0, step=0, sample_offset=0):
    """-> dict of arrays: rir (int, -1 = none), noise (int, -1 = none), offset (int), snr_db (float32)."""
    g = (np.arange(B, dtype=np.uint64) + np.uint64(sample_offset))
    key = make_key(seed)
    r_rir = philox4x32_10(make_ctr(step, g, TAG_AUDIO, 0), key).astype(np.uint64)
    r_noi = philox4x32_10(make_ctr(step, g, TAG_AUDIO, 1), key).astype(np.uint64)
    rir = np.full(B, -1, dtype=np.int64)
    if R > 0:
        app = r_rir[:, 0] < np.uint64(prob_threshold(rir_prob))
        rir = np.where(app, (r_rir[:, 1] % np.uint64(R)).astype(np.int64), -1)
    noise = np.full(B, -1, dtype=np.int64)
    offset = np.zeros(B, dtype=np.int64)
    if K > 0:
        app = r_noi[:, 0] < np.uint64(prob_threshold(noise_prob))
        noise = np.where(app, (r_noi[:, 1] % np.uint64(K)).astype(np.int64), -1)
        offset = (r_noi[:, 2] % np.uint64(Nn - N + 1)).astype(np.int64)
    u = (r_noi[:, 3] >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)
    snr = (u.astype(np.float64) * float(np.float32(snr_max) - np.float32(snr_min)) + float(np.float32(snr_min))).astype(np.float32)
    return dict(rir=rir, noise=noise, offset=offset, snr_db=snr)


def audio_augment(x, rirs, noises, rir_prob, noise_prob, snr_min, snr_max, seed=0, step=0, sample_offset=0):
    """x (B,N) -> (out (B,N) float64, choices)."""
    x = np.asarray(x, dtype=np.float64)
    B, N = x.shape
    R = 0 if rirs is None else rirs.shape[0]
    K, Nn = (0, 0) if noises is None else noises.shape
    ch = audio_choices(B, N, R, K, Nn, rir_prob, noise_prob, snr_min, snr_max, seed, step, sample_offset)
    out = np.empty_like(x)
    for b in range(B):
        y = x[b]
        rx = np.sqrt(np.mean(y ** 2))
        if ch["rir"][b] >= 0:
            y = np.convolve(x[b], np.asarray(rirs[ch["rir"][b]], dtype=np.float64))[:N]
            ry = np.sqrt(np.mean(y ** 2))
            if ry > 0:
                y = y * (rx / ry)
        if ch["noise"][b] >= 0:
            n = np.asarray(noises[ch["noise"][b]], dtype=np.float64)[ch["offset"][b]:ch["offset"][b] + N]
            rn = np.sqrt(np.mean(n ** 2))
            ry = np.sqrt(np.mean(y ** 2))
            gain = ry / (rn * 10.0 ** (float(ch["snr_db"][b]) / 20.0)) if rn > 0 else 0.0
            y = y + gain * n
        out[b] = np.clip(y, -1.0, 1.0)
    return out, ch
