"""Waveform augmentation: oracle self-checks on CPU, HIP parity on the GPU.
The law is the build's own spec (reference source absent -> parity unpinned w.r.t. the reference); what is checked here
is device == oracle: Philox choices bit-exact, waveform within 1e-4 absolute (fp32 direct convolution vs float64)."""
import numpy as np
import pytest
import torch

from oracle import audio_augment as oa


def _banks(rng, R=3, L=1200, K=2, Nn=40000):
    t = np.arange(L)
    rirs = (rng.standard_normal((R, L)) * np.exp(-t / (L / 6.0))).astype(np.float32)
    rirs[:, 0] = 1.0
    noises = (0.1 * rng.standard_normal((K, Nn))).astype(np.float32)
    return rirs, noises


def test_oracle_choices_ranges_and_rates():
    ch = oa.audio_choices(4096, 24000, 5, 7, 48000, 0.25, 0.5, 5.0, 20.0, seed=3, step=11)
    assert ch["rir"].max() < 5 and ch["noise"].max() < 7
    assert 0 <= ch["offset"].min() and ch["offset"].max() <= 24000
    assert ch["snr_db"].min() >= 5.0 and ch["snr_db"].max() <= 20.0
    assert abs((ch["rir"] >= 0).mean() - 0.25) < 0.03 and abs((ch["noise"] >= 0).mean() - 0.5) < 0.03
    # a different step or sample offset is a different stream; same arguments repeat exactly
    ch2 = oa.audio_choices(4096, 24000, 5, 7, 48000, 0.25, 0.5, 5.0, 20.0, seed=3, step=12)
    assert (ch2["offset"] != ch["offset"]).mean() > 0.9
    ch3 = oa.audio_choices(8, 24000, 5, 7, 48000, 0.25, 0.5, 5.0, 20.0, seed=3, step=11, sample_offset=100)
    assert np.array_equal(ch3["offset"], ch["offset"][100:108])


def test_oracle_signal_law():
    rng = np.random.default_rng(0)
    rirs, noises = _banks(rng)
    x = (0.2 * rng.standard_normal((16, 8000))).astype(np.float32)
    out, ch = oa.audio_augment(x, rirs, noises, 1.0, 1.0, 10.0, 10.0, seed=1)
    assert out.shape == x.shape and np.isfinite(out).all() and np.abs(out).max() <= 1.0
    for b in range(16):       # SNR of the mix is the drawn one (no clipping at this level)
        n = noises[ch["noise"][b]][ch["offset"][b]:ch["offset"][b] + 8000].astype(np.float64)
        y = np.convolve(x[b].astype(np.float64), rirs[ch["rir"][b]].astype(np.float64))[:8000]
        y *= np.sqrt(np.mean(x[b].astype(np.float64) ** 2) / np.mean(y ** 2))
        resid = out[b] - y
        snr = 10 * np.log10(np.mean(y ** 2) / np.mean(resid ** 2))
        assert abs(snr - 10.0) < 1e-6
        assert np.allclose(resid / np.sqrt(np.mean(resid ** 2)), n / np.sqrt(np.mean(n ** 2)), atol=1e-9)
    # probabilities 0 -> identity (up to the clip)
    out0, _ = oa.audio_augment(x, rirs, noises, 0.0, 0.0, 5.0, 20.0)
    assert np.array_equal(out0, np.clip(x.astype(np.float64), -1, 1))
    # no banks -> identity
    out1, ch1 = oa.audio_augment(x, None, None, 1.0, 1.0, 5.0, 20.0)
    assert np.array_equal(out1, out0) and (ch1["rir"] == -1).all() and (ch1["noise"] == -1).all()


# ---------------------------------------------------------------------------------------------------- GPU parity
def _run_device(x, rirs, noises, fft=False, **kw):
    from wakeword_trainer_home_amd import _native as nat
    dev = torch.device("cuda:0")
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    spectra = nat.audio_rir_spectra(t(rirs)) if (fft and rirs is not None) else None
    out, ch = nat.audio_augment(t(x), t(rirs), t(noises), want_choices=True, rir_spectra=spectra, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy(), ch.cpu().numpy()


def _check(x, rirs, noises, rir_prob, noise_prob, smin, smax, seed=0, step=0, sample_offset=0, atol=1e-4, fft=False):
    out, ch = _run_device(x, rirs, noises, fft=fft, rir_prob=rir_prob, noise_prob=noise_prob, snr_min_db=smin, snr_max_db=smax,
                          seed=seed, step=step, sample_offset=sample_offset)
    ref, rch = oa.audio_augment(x, rirs, noises, rir_prob, noise_prob, smin, smax, seed, step, sample_offset)
    assert np.array_equal(ch[:, 0], rch["rir"]) and np.array_equal(ch[:, 1], rch["noise"])
    assert np.array_equal(ch[:, 2], rch["offset"])
    assert np.allclose(ch[:, 3].copy().view(np.float32), rch["snr_db"], rtol=0, atol=2e-6)
    assert np.isfinite(out).all() and np.abs(out).max() <= 1.0
    err = np.abs(out - ref).max()
    assert err <= atol, err
    return out, ch


@pytest.mark.gpu
@pytest.mark.parametrize("fft", [False, True], ids=["direct", "fft"])
@pytest.mark.parametrize("B,N,L", [(6, 24000, 1200), (3, 2048, 8), (5, 5000, 1), (2, 24000, 8192), (4, 1000, 3001),
                                   (3, 24000, 4000), (2, 40000, 5000)])
def test_device_matches_oracle(B, N, L, fft):
    """Both forms of the convolution: direct time-domain, and overlap-save FFT (1, 2 and 3+ segments per clip)."""
    rng = np.random.default_rng(B * 1000 + L)
    rirs, noises = _banks(rng, R=3, L=L, K=2, Nn=N + 777)
    x = (0.2 * rng.standard_normal((B, N))).astype(np.float32)
    _check(x, rirs, noises, 0.6, 0.6, 5.0, 20.0, seed=5, step=3, sample_offset=17, fft=fft)
    _check(x, rirs, noises, 1.0, 1.0, 0.0, 0.0, seed=6, fft=fft)


@pytest.mark.gpu
def test_fft_and_direct_forms_agree_at_full_batch():
    """BASELINE-sized batch (512 clips x 24000 samples, 4000-tap RIRs): the two device forms against each other
    (the float64 oracle covers the small cases above)."""
    from wakeword_trainer_home_amd import _native as nat
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    x = 0.2 * torch.randn(512, 24000, device=dev, generator=g)
    rirs = torch.randn(16, 4000, device=dev, generator=g) * torch.exp(-torch.arange(4000, device=dev) / 700.0)
    noises = 0.1 * torch.randn(8, 160000, device=dev, generator=g)
    kw = dict(rir_prob=0.25, noise_prob=0.5, snr_min_db=5.0, snr_max_db=20.0, seed=1, step=9, want_choices=True)
    a, ca = nat.audio_augment(x, rirs, noises, **kw)
    b, cb = nat.audio_augment(x, rirs, noises, rir_spectra=nat.audio_rir_spectra(rirs), **kw)
    assert torch.equal(ca, cb)
    assert 0.15 < (ca[:, 0] >= 0).float().mean() < 0.35 and 0.4 < (ca[:, 1] >= 0).float().mean() < 0.6
    assert torch.isfinite(b).all() and (a - b).abs().max().item() <= 1e-4
    plain = (ca[:, 0] < 0) & (ca[:, 1] < 0)                       # untouched clips are bit-exact copies
    assert torch.equal(b[plain], x[plain].clamp(-1, 1))


@pytest.mark.gpu
def test_device_edge_cases():
    rng = np.random.default_rng(9)
    rirs, noises = _banks(rng, R=2, L=400, K=3, Nn=24000)       # Nn == N: the only offset is 0
    x = (0.3 * rng.standard_normal((8, 24000))).astype(np.float32)
    out, ch = _check(x, rirs, noises, 0.5, 0.5, 5.0, 20.0, seed=2)
    assert (ch[:, 2] == 0).all()
    # effects off / banks absent: clipped copy, bit-exact
    for args in ((rirs, noises, 0.0, 0.0), (None, None, 1.0, 1.0)):
        o, c = _run_device(x * 5, args[0], args[1], rir_prob=args[2], noise_prob=args[3], snr_min_db=5.0, snr_max_db=20.0)
        assert np.array_equal(o, np.clip(x * 5, -1, 1)) and (c[:, :2] == -1).all()
    # silent clip and silent noise: no NaN from 0/0
    x[0] = 0
    noises[:] = 0
    _check(x, rirs, noises, 1.0, 1.0, 5.0, 20.0, seed=4)
    _check(x, rirs, noises, 1.0, 1.0, 5.0, 20.0, seed=4, fft=True)
    # only one of the banks
    _check(x, rirs, None, 1.0, 1.0, 5.0, 20.0, seed=4)
    _check(x, rirs, None, 1.0, 1.0, 5.0, 20.0, seed=4, fft=True)
    _check(x, None, _banks(rng, K=2, Nn=30000)[1], 1.0, 1.0, 5.0, 20.0, seed=4)


@pytest.mark.gpu
def test_device_rejects_bad_arguments():
    from wakeword_trainer_home_amd import _native as nat
    dev = torch.device("cuda:0")
    x = torch.zeros(2, 4000, device=dev)
    with pytest.raises(nat.NativeError):
        nat.audio_augment(x, torch.zeros(1, 9000, device=dev), None, 1.0, 0.0, 5.0, 20.0)     # RIR too long
    with pytest.raises(ValueError):
        nat.audio_augment(x, None, torch.zeros(1, 3999, device=dev), 0.0, 1.0, 5.0, 20.0)     # noise shorter than N
    with pytest.raises(ValueError):
        nat.audio_augment(x, None, None, 0.0, 0.0, 20.0, 5.0)                                 # snr range reversed
    with pytest.raises(ValueError):
        nat.audio_augment(x.double(), None, None, 0.0, 0.0, 5.0, 20.0)


@pytest.mark.gpu
def test_audio_augmentation_class_and_trainer_hook(tmp_path):
    """Constructor and call contract of tests/test_training_pipeline.py:230-243, then the Trainer applies it ahead of the
    log-mel with the step / sample-offset counters of the batch."""
    from wakeword_trainer_home_amd.data.augmentation import AudioAugmentation
    from wakeword_trainer_home_amd.data.dataset import make_synthetic_batch
    from oracle import features as of
    rng = np.random.default_rng(1)
    aug = AudioAugmentation(sample_rate=16000, device="cuda", time_stretch_range=(0.8, 1.2), pitch_shift_range=(-2, 2),
                            background_noise_prob=0.5)
    a = torch.randn(1, 16000, device="cuda")
    o = aug(a)
    assert o.shape == a.shape and torch.isfinite(o).all()
    rirs, noises = _banks(rng, R=4, L=800, K=3, Nn=30000)
    aug = AudioAugmentation(device="cuda", background_noise_prob=0.7, rir_prob=0.5, rirs=rirs, noises=noises, seed=9)
    x = (0.2 * rng.standard_normal((8, 24000))).astype(np.float32)
    o0 = aug(torch.from_numpy(x), return_choices=True).cpu().numpy()
    ref0, rch = oa.audio_augment(x, rirs, noises, 0.5, 0.7, 5.0, 20.0, seed=9, step=0)
    assert np.array_equal(aug.last_choices[:, 0].cpu().numpy(), rch["rir"]) and np.abs(o0 - ref0).max() <= 1e-4
    o1 = aug(torch.from_numpy(x)).cpu().numpy()                      # the call counter advanced the stream
    ref1, _ = oa.audio_augment(x, rirs, noises, 0.5, 0.7, 5.0, 20.0, seed=9, step=1)
    assert np.abs(o1 - ref1).max() <= 1e-4 and np.abs(o1 - o0).max() > 1e-3
    xi = (x * 32767).astype(np.int16)                                # int16 PCM input
    oi = aug(torch.from_numpy(xi), step=0).cpu().numpy()
    refi, _ = oa.audio_augment(xi.astype(np.float32) / 32768.0, rirs, noises, 0.5, 0.7, 5.0, 20.0, seed=9, step=0)
    assert np.abs(oi - refi).max() <= 1e-4

    # Trainer hook: features of step 2 == log-mel(waveform augmented with step=2, offset=rank*B); SpecAugment off
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    cfg.augmentation.n_freq_masks = cfg.augmentation.n_time_masks = 0
    y = torch.zeros(8, dtype=torch.long)
    batches = [(torch.from_numpy(x), y, [{"path": "s"}] * 8)]
    tr = Trainer(create_model("cnn_small"), batches, batches, cfg, checkpoint_dir=tmp_path, device="cuda")
    tr.audio_augmentation = aug
    feats = tr._features(torch.from_numpy(x), training=True, step=2).cpu().numpy()
    w2 = aug(torch.from_numpy(x), step=2).cpu().numpy()
    ref2, _ = oa.audio_augment(x, rirs, noises, 0.5, 0.7, 5.0, 20.0, seed=9, step=2)
    assert np.abs(w2 - ref2).max() <= 1e-4
    lm = of.logmel(w2)
    assert np.abs(feats.reshape(lm.shape) - lm).max() <= 2e-3
    ev = tr._features(torch.from_numpy(x), training=False).cpu().numpy()      # evaluation: no augmentation
    assert np.abs(ev.reshape(lm.shape) - of.logmel(x)).max() <= 2e-3
    tr.train_epoch(0)                                                          # and a full step runs through it
    assert np.isfinite(tr.train_metrics_tracker.compute().accuracy)
