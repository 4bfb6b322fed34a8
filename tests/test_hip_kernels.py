"""GPU parity tests: every HIP kernel, called through the C-ABI, against the CPU oracle
(float64 numpy / torch on the host) on the same seeded inputs.

Tolerances (fp32 kernels): features 1e-3 absolute on log-mel (north_star), conv layers
2e-4 relative to the tensor's scale, SpecAugment / dropout indices bit-exact.
"""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wakeword_trainer_home_amd import _native
    _native.load()
    return _native


DEV = "cuda:0"


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(DEV).contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


# --------------------------------------------------------------------------- RNG / SpecAugment
def test_host_philox_and_threshold_match_oracle(nat):
    from oracle.philox import philox4x32_10, prob_threshold
    rng = np.random.default_rng(0)
    for _ in range(20):
        c = rng.integers(0, 2 ** 32, 4, dtype=np.uint64)
        k = rng.integers(0, 2 ** 32, 2, dtype=np.uint64)
        assert nat.philox([int(v) for v in c], [int(v) for v in k]) == [int(v) for v in philox4x32_10(c, k)]
    for p in (0.0, 0.2, 0.3, 0.5, 0.999, 1.0):
        assert nat.prob_threshold(float(np.float32(p))) == prob_threshold(p)


@pytest.mark.parametrize("shape,cfg", [
    ((64, 40, 151), dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2,
                         freq_mask_prob=1.0, time_mask_prob=1.0)),
    ((33, 40, 151), dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2,
                         freq_mask_prob=0.5, time_mask_prob=0.5)),
    ((5, 64, 50), dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2,
                       freq_mask_prob=0.2, time_mask_prob=0.2)),
    ((7, 8, 10), dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=3, n_time_masks=1,
                      freq_mask_prob=1.0, time_mask_prob=1.0)),
    ((4, 40, 151), dict(freq_mask_param=27, time_mask_param=100, n_freq_masks=0, n_time_masks=4,
                        freq_mask_prob=1.0, time_mask_prob=0.7)),
])
def test_specaug_indices_bit_exact(nat, shape, cfg):
    from oracle.specaugment import specaug_indices, specaug_apply
    B, Fd, T = shape
    x = np.random.default_rng(1).normal(size=(B, 1, Fd, T)).astype(np.float32)
    for seed, step, off in ((2024, 0, 0), (2 ** 40 + 17, 2 ** 33 + 5, 1000)):
        xg = cu(x)
        idx = nat.specaug_apply_(xg, nat.make_specaug_cfg(**cfg), seed=seed, step=step, sample_offset=off,
                                 want_idx=True)
        ref_idx = specaug_indices(B, Fd, T, seed=seed, step=step, sample_offset=off, **cfg)
        assert np.array_equal(idx.cpu().numpy(), ref_idx)
        assert np.array_equal(xg.cpu().numpy(), specaug_apply(x, ref_idx, cfg["n_freq_masks"]))


# --------------------------------------------------------------------------- features
def _waves(B, N, seed=0):
    rng = np.random.default_rng(seed)
    x = np.clip(rng.normal(0, 0.1, (B, N)), -1, 1).astype(np.float32)
    t = np.arange(N) / 16000.0
    if B > 1:
        x[1] = 0.5 * np.sin(2 * np.pi * (200 + 3000 * t) * t)        # sweep
    if B > 2:
        x[2] = 0.0                                                    # silence
    if B > 3:
        x[3] = np.sign(np.sin(2 * np.pi * 440 * t))                   # full-scale square
    return x


def _assert_logmel_close(out, ref, x, tol=1e-3, **kw):
    """north_star tolerance: 1e-3 absolute on log-mel, per clip.  Stated exception: a clip whose spectrum
    spans > 100 dB (the synthetic pure sweep, clip 1 of _waves) sits on the fp32 round-off floor of ANY fp32
    STFT under log(mel + 1e-6) -- torch.stft in fp32 is itself 1.3-1.6e-3 off the float64 oracle there -- so
    for such a clip the bound is 2x the error of the fp32 torch.stft formulation of the same spec."""
    from oracle import features as OF
    err = np.abs(out - ref).reshape(out.shape[0], -1).max(axis=1)
    t32 = np.abs(OF.logmel_torch(x, **kw).numpy() - ref).reshape(out.shape[0], -1).max(axis=1)
    for b, (e, t) in enumerate(zip(err, t32)):
        bound = tol if t < 0.5 * tol else max(tol, 2.0 * t)
        assert e < bound, f"clip {b}: log-mel max abs err {e:.3e} (bound {bound:.1e}, torch fp32 {t:.1e})"


@pytest.mark.parametrize("B,N,kw", [
    (6, 24000, dict()),
    (3, 24000, dict(n_mels=128)),
    (4, 16000, dict(hop=256, n_mels=64)),
    (2, 40000, dict()),
    (5, 600, dict()),
    (3, 24000, dict(f_min=50.0, f_max=7600.0)),
    # band counts that are not a multiple of the MFMA band sums' 4-band blocks / one block only / three passes' worth
    (2, 24000, dict(n_mels=13)),
    (2, 8000, dict(n_mels=3)),
    (2, 24000, dict(n_mels=80, f_min=20.0)),
    (2, 24000, dict(n_mels=23, f_max=3800.0)),
])
def test_logmel_matches_oracle(nat, B, N, kw):
    from oracle import features as OF
    x = _waves(B, N)
    n_mels, hop = kw.get("n_mels", 40), kw.get("hop", 160)
    cfg = nat.make_feat_cfg(n_mels=n_mels, hop=hop, f_min=kw.get("f_min", 0.0), f_max=kw.get("f_max", 0.0))
    out = nat.logmel_fwd(cu(x), cfg).cpu().numpy()
    ref = OF.logmel(x, hop=hop, n_mels=n_mels, f_min=kw.get("f_min", 0.0), f_max=kw.get("f_max") or None)
    assert out.shape == ref.shape == (B, 1, n_mels, 1 + N // hop)
    _assert_logmel_close(out, ref, x, hop=hop, n_mels=n_mels, f_min=kw.get("f_min", 0.0), f_max=kw.get("f_max") or None)


@pytest.mark.parametrize("n_fft,hop,n_mels,N", [(256, 64, 40, 8000), (512, 160, 40, 24000), (2048, 512, 64, 24000),
                                                (4096, 160, 40, 24000), (512, 128, 128, 5000), (64, 32, 13, 3000),
                                                (128, 160, 23, 24000), (256, 160, 40, 200), (512, 160, 80, 300)])
def test_logmel_other_fft_sizes_match_oracle(nat, n_fft, hop, n_mels, N):
    """n_fft other than the reference default 1024 (its validator accepts 256 ... 4096, src/config/validator.py:129): below 1024
    the 1024-point kernel on zero-extended frames (clips shorter than its 512-sample reach included), above it the general
    radix-2 kernel; same spec and bound; also int16 input, MFCC and the fused SpecAugment on those paths."""
    from oracle import features as OF
    from oracle.specaugment import specaug_indices, specaug_apply
    x = _waves(4, N, seed=n_fft)
    cfg = nat.make_feat_cfg(n_fft=n_fft, hop=hop, n_mels=n_mels)
    out = nat.logmel_fwd(cu(x), cfg).cpu().numpy()
    ref = OF.logmel(x, n_fft=n_fft, hop=hop, n_mels=n_mels)
    assert out.shape == ref.shape == (4, 1, n_mels, 1 + N // hop)
    _assert_logmel_close(out, ref, x, n_fft=n_fft, hop=hop, n_mels=n_mels)
    xi = np.round(x * 32767).astype(np.int16)
    oi = nat.logmel_fwd(cu(xi, torch.int16), cfg).cpu().numpy()
    _assert_logmel_close(oi, OF.logmel(xi.astype(np.float64) / 32768.0, n_fft=n_fft, hop=hop, n_mels=n_mels),
                         xi.astype(np.float32) / 32768.0, n_fft=n_fft, hop=hop, n_mels=n_mels)
    mf = nat.logmel_fwd(cu(x), nat.make_feat_cfg(n_fft=n_fft, hop=hop, n_mels=n_mels, n_mfcc=13)).cpu().numpy()
    assert np.abs(mf - OF.mfcc(x, n_fft=n_fft, hop=hop, n_mels=n_mels, n_mfcc=13)).max() < 3e-3
    T = 1 + N // hop
    sa = dict(freq_mask_param=15, time_mask_param=min(35, T), n_freq_masks=2, n_time_masks=2, freq_mask_prob=0.7, time_mask_prob=0.7)
    fused, idx = nat.logmel_fwd(cu(x), cfg, nat.make_specaug_cfg(**sa), seed=3, step=9, sample_offset=5, want_idx=True)
    ridx = specaug_indices(4, n_mels, T, seed=3, step=9, sample_offset=5, **sa)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(fused.cpu().numpy(), specaug_apply(out, ridx, 2))


def test_logmel_int16_and_mfcc_and_fused_specaug(nat):
    from oracle import features as OF
    from oracle.specaugment import specaug_indices, specaug_apply
    x = _waves(5, 24000, seed=3)
    xi = np.round(x * 32767).astype(np.int16)
    out = nat.logmel_fwd(cu(xi, torch.int16), nat.make_feat_cfg()).cpu().numpy()
    ref = OF.logmel(xi.astype(np.float64) / 32768.0)
    _assert_logmel_close(out, ref, xi.astype(np.float32) / 32768.0)
    mf = nat.logmel_fwd(cu(x), nat.make_feat_cfg(n_mfcc=13)).cpu().numpy()
    ref_mf = OF.mfcc(x, n_mfcc=13)
    assert mf.shape == (5, 1, 13, 151)
    assert np.abs(mf - ref_mf).max() < 2e-3
    sa = dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2, freq_mask_prob=0.5,
              time_mask_prob=0.5)
    fused, idx = nat.logmel_fwd(cu(x), nat.make_feat_cfg(), nat.make_specaug_cfg(**sa), seed=7, step=3,
                                sample_offset=11, want_idx=True)
    ridx = specaug_indices(5, 40, 151, seed=7, step=3, sample_offset=11, **sa)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    plain = nat.logmel_fwd(cu(x), nat.make_feat_cfg()).cpu().numpy()
    assert np.array_equal(fused.cpu().numpy(), specaug_apply(plain, ridx, 2))


def test_logmel_persistent_grid_size_does_not_change_results(nat):
    """ww_ctx_set_logmel_workgroups is a launch parameter: one workgroup walking every (clip, frame block) item, three, one
    per item and the full-device default all give the same bits -- features, MFCC and the SpecAugment rows (which only the
    first block of a clip reports)."""
    x = cu(_waves(7, 24000, seed=21))
    xi = cu(np.round(_waves(3, 9000, seed=22) * 32767).astype(np.int16), torch.int16)
    sa = nat.make_specaug_cfg(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2, freq_mask_prob=0.6,
                              time_mask_prob=0.6)

    def run():
        f, idx = nat.logmel_fwd(x, nat.make_feat_cfg(), sa, seed=5, step=2, sample_offset=40, want_idx=True)
        return f, idx, nat.logmel_fwd(x, nat.make_feat_cfg(n_mfcc=13)), nat.logmel_fwd(xi, nat.make_feat_cfg(n_mels=64))

    ref = run()
    try:
        for n in (1, 3, 64, 70, 100000):
            nat.set_logmel_workgroups("cuda:0", n)
            for a, b in zip(run(), ref):
                assert torch.equal(a, b), n
        with pytest.raises(ValueError):
            nat.set_logmel_workgroups("cuda:0", -1)
    finally:
        nat.set_logmel_workgroups("cuda:0", 0)


def test_logmel_rejects_bad_input(nat):
    with pytest.raises(ValueError):
        nat.logmel_fwd(cu(np.zeros((2, 100), np.float32)), nat.make_feat_cfg())      # N <= n_fft/2
    with pytest.raises(nat.NativeError):
        nat.logmel_fwd(cu(np.zeros((2, 4000), np.float32)), nat.make_feat_cfg(n_fft=1000))     # not a power of two
    with pytest.raises(nat.NativeError):
        nat.logmel_fwd(cu(np.zeros((2, 40000), np.float32)), nat.make_feat_cfg(n_fft=8192))   # beyond 4096
    with pytest.raises(nat.NativeError):
        nat.logmel_fwd(torch.zeros(2, 4000), nat.make_feat_cfg())                    # CPU tensor: no fallback


# --------------------------------------------------------------------------- forward layers
def _bn_tensors(seed):
    g = torch.Generator().manual_seed(seed)
    gamma = torch.rand(64, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(64, generator=g, dtype=torch.float64) * 0.3
    rm = torch.randn(64, generator=g, dtype=torch.float64) * 0.1
    rv = torch.rand(64, generator=g, dtype=torch.float64) + 0.5
    return gamma, beta, rm, rv


def _check_bn_outputs(y_ref_nchw, gamma, beta, rm, rv, ss, mr, rm_new, rv_new, mom=0.1, eps=1e-5):
    mean = y_ref_nchw.mean(dim=(0, 2, 3))
    var = y_ref_nchw.var(dim=(0, 2, 3), unbiased=False)
    n = y_ref_nchw.numel() / 64
    rstd = 1.0 / torch.sqrt(var + eps)
    scale = gamma * rstd
    shift = beta - mean * scale
    assert rel_err(ss[:64].cpu(), scale) < 2e-5
    assert np.abs(ss[64:].cpu().numpy() - shift.numpy()).max() < 2e-5 * (shift.abs().max().item() + 1)
    assert np.abs(mr[:64].cpu().numpy() - mean.numpy()).max() < 1e-5 * (mean.abs().max().item() + 1)
    assert rel_err(mr[64:].cpu(), rstd) < 2e-5
    assert np.abs(rm_new.cpu().numpy() - ((1 - mom) * rm + mom * mean).numpy()).max() < 1e-5
    assert rel_err(rv_new.cpu(), (1 - mom) * rv + mom * var * n / (n - 1)) < 2e-5


@pytest.mark.parametrize("B,Hin,Win", [(3, 40, 151), (2, 13, 50), (1, 64, 50), (2, 7, 9)])
def test_stem_fwd(nat, B, Hin, Win):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(B, 1, Hin, Win, generator=g, dtype=torch.float64) * 2 - 4
    w = torch.randn(64, 1, 3, 3, generator=g, dtype=torch.float64) * 0.3
    gamma, beta, rm, rv = _bn_tensors(1)
    ref = F.conv2d(x, w, stride=2, padding=1)
    rm_g, rv_g = cu(rm), cu(rv)
    ga, be = cu(gamma), cu(beta)          # keep the BN tensors alive for the duration of the call
    bn = nat.make_bn(ga, be, rm_g, rv_g)
    y, ss, mr = nat.conv_stem_fwd(cu(x), cu(w), bn, nat.layer_scratch(DEV))
    assert y.shape == (B, (Hin + 1) // 2, (Win + 1) // 2, 64)
    assert rel_err(y.cpu(), nhwc(ref)) < 1e-5
    _check_bn_outputs(ref, gamma, beta, rm, rv, ss, mr, rm_g, rv_g)


@pytest.mark.parametrize("kind", ["dw", "pw"])
@pytest.mark.parametrize("B,H,W", [(3, 20, 76), (2, 7, 25), (1, 5, 3), (5, 17, 10)])
def test_conv_fwd_layers(nat, kind, B, H, W):
    g = torch.Generator().manual_seed(11)
    y_in = torch.randn(B, 64, H, W, generator=g, dtype=torch.float64)
    s_in = torch.rand(64, generator=g, dtype=torch.float64) + 0.5
    t_in = torch.randn(64, generator=g, dtype=torch.float64) * 0.5
    a = torch.relu(y_in * s_in[None, :, None, None] + t_in[None, :, None, None])
    if kind == "dw":
        w = torch.randn(64, 1, 3, 3, generator=g, dtype=torch.float64) * 0.3
        ref = F.conv2d(a, w, padding=1, groups=64)
    else:
        w = torch.randn(64, 64, 1, 1, generator=g, dtype=torch.float64) * 0.2
        ref = F.conv2d(a, w)
    gamma, beta, rm, rv = _bn_tensors(2)
    ga, be, rm_g, rv_g = cu(gamma), cu(beta), cu(rm), cu(rv)
    bn = nat.make_bn(ga, be, rm_g, rv_g)
    fn = nat.dwconv3x3_fwd if kind == "dw" else nat.pwconv1x1_fwd
    ss_in = cu(torch.cat([s_in, t_in]))
    y, ss, mr = fn(cu(nhwc(y_in)), ss_in, cu(w), bn, nat.layer_scratch(DEV))
    assert rel_err(y.cpu(), nhwc(ref)) < 2e-5, kind
    _check_bn_outputs(ref, gamma, beta, rm, rv, ss, mr, rm_g, rv_g)
    # eval mode: scale/shift from running statistics, running statistics untouched
    bn_e = nat.make_bn(ga, be, rm_g, rv_g, training=False)
    rm_before, rv_before = rm_g.clone(), rv_g.clone()
    y2, ss2, _ = fn(cu(nhwc(y_in)), ss_in, cu(w), bn_e, nat.layer_scratch(DEV))
    assert torch.equal(y2, y)
    assert torch.equal(rm_g, rm_before) and torch.equal(rv_g, rv_before)
    sc = gamma / torch.sqrt(rv_g.cpu().double() + 1e-5)
    assert rel_err(ss2[:64].cpu(), sc) < 1e-5


def test_gap_and_head_fwd(nat):
    from oracle.cnn_small import dropout_keep_mask
    g = torch.Generator().manual_seed(12)
    B, H, W = 9, 20, 76
    y = torch.randn(B, 64, H, W, generator=g, dtype=torch.float64)
    s = torch.rand(64, generator=g, dtype=torch.float64) + 0.5
    t = torch.randn(64, generator=g, dtype=torch.float64) * 0.5
    mean = torch.randn(64, generator=g, dtype=torch.float64) * 0.2
    rstd = torch.rand(64, generator=g, dtype=torch.float64) + 0.7
    z = y * s[None, :, None, None] + t[None, :, None, None]
    pos = z > 0
    yhat = (y - mean[None, :, None, None]) * rstd[None, :, None, None]
    pool = nat.gap_fwd(cu(nhwc(y)), cu(torch.cat([s, t])), cu(torch.cat([mean, rstd])))
    assert rel_err(pool[:, 0].cpu(), torch.relu(z).sum(dim=(2, 3))) < 1e-5
    assert np.abs(pool[:, 1].cpu().numpy() - (yhat * pos).sum(dim=(2, 3)).numpy()).max() < 2e-3
    # an fp32 z within 1 ulp of 0 may land on the other side of the ReLU than the float64 reference
    assert (pool[:, 2].cpu().double() - pos.sum(dim=(2, 3)).double()).abs().max() <= 2
    fc_w = torch.randn(2, 64, generator=g, dtype=torch.float64) * 0.2
    fc_b = torch.randn(2, generator=g, dtype=torch.float64)
    pooled = torch.relu(z).mean(dim=(2, 3))
    for p, training in ((0.0, True), (0.3, True), (0.3, False)):
        pd, logits = nat.head_fwd(pool, H * W, cu(fc_w), cu(fc_b), dropout_p=p, training=training, seed=5, step=9,
                                  sample_offset=3)
        if training and p > 0:
            keep = dropout_keep_mask(B, 64, p, seed=5, step=9, sample_offset=3)
            assert np.array_equal(pd.cpu().numpy() != 0, keep & (pooled.numpy() != 0))
            ref_pd = pooled * torch.from_numpy(keep.astype(np.float64)) / (1.0 - float(np.float32(p)))
        else:
            ref_pd = pooled
        assert rel_err(pd.cpu(), ref_pd) < 1e-5
        assert rel_err(logits.cpu(), ref_pd @ fc_w.t() + fc_b) < 1e-5


# --------------------------------------------------------------------------- loss
def test_loss_matches_reference_fixture(nat, golden_dir):
    z = np.load(golden_dir / "g1_loss.npz")
    specs = json.loads(str(z["specs"]))
    for case in ("b512", "b7", "extreme"):
        logits, targets = cu(z[f"{case}/logits"]), cu(z[f"{case}/targets"], torch.int64)
        for i, (name, kw) in enumerate(specs):
            kind = nat.LOSS_CE if name == "cross_entropy" else nat.LOSS_FOCAL
            loss, dl, stats = nat.ce2_loss_fwd_bwd(logits, targets, kind, kw.get("label_smoothing", 0.0),
                                                   kw.get("focal_alpha", 0.25), kw.get("focal_gamma", 2.0))
            ref_loss = float(z[f"{case}/spec{i}/loss"])
            assert abs(loss.item() - ref_loss) < 2e-6 * max(1.0, abs(ref_loss)), (case, name, kw)
            assert np.abs(dl.cpu().numpy() - z[f"{case}/spec{i}/dlogits"]).max() < 2e-7, (case, name, kw)
            st = nat.decode_stats(stats.cpu())
            assert st["nonfinite"] == 0 and st["bad_target"] == 0 and st["count"] == logits.shape[0]
            assert abs(st["loss"] - ref_loss) < 2e-6 * max(1.0, abs(ref_loss))


def test_loss_counters_and_flags(nat, golden_dir):
    from oracle.losses import batch_counters
    cases = json.loads((golden_dir / "g3_metrics.json").read_text())
    for name, c in cases.items():
        logits, targets = np.array(c["logits"], np.float32), np.array(c["targets"])
        _, _, stats = nat.ce2_loss_fwd_bwd(cu(logits), cu(targets, torch.int64), nat.LOSS_CE, 0.05)
        st = nat.decode_stats(stats.cpu())
        correct, tp, tn, fp, fn = batch_counters(logits, targets)
        assert (st["correct"], st["tp"], st["tn"], st["fp"], st["fn"]) == (correct, tp, tn, fp, fn), name
        r = c["result"]
        assert (st["tp"], st["tn"], st["fp"], st["fn"]) == (r["true_positives"], r["true_negatives"],
                                                            r["false_positives"], r["false_negatives"])
    bad = cu(np.array([0, 2, 1, -1]), torch.int64)
    _, _, stats = nat.ce2_loss_fwd_bwd(cu(np.zeros((4, 2), np.float32)), bad)
    assert nat.decode_stats(stats.cpu())["bad_target"] == 1
    inf = cu(np.array([[np.inf, 0.0], [0.0, 1.0]], np.float32))
    _, _, stats = nat.ce2_loss_fwd_bwd(inf, cu(np.array([1, 0]), torch.int64))
    assert nat.decode_stats(stats.cpu())["nonfinite"] == 1
    with pytest.raises(ValueError):
        nat.ce2_loss_fwd_bwd(cu(np.zeros((4, 2), np.float32)), cu(np.zeros(4), torch.int64), label_smoothing=1.5)


def test_grad_norm_clip(nat):
    g = torch.Generator().manual_seed(13)
    v = torch.randn(20546, generator=g) * 0.3
    for max_norm in (1.0, 1e6, 0.0):
        flat = v.clone().to(DEV)
        norm = nat.grad_norm_clip_(flat, max_norm)
        ref = v.clone().requires_grad_(False)
        p = torch.nn.Parameter(torch.zeros_like(ref))
        p.grad = ref.clone()
        if max_norm > 0:
            ref_norm = torch.nn.utils.clip_grad_norm_([p], max_norm)
        else:
            ref_norm = ref.norm()
        assert abs(norm.item() - ref_norm.item()) < 1e-4 * ref_norm.item()
        assert rel_err(flat.cpu(), p.grad) < 1e-5


# --------------------------------------------------------------------------- backward layers
def _coef_from(g_nchw, y_nchw, gamma, eps=1e-5):
    """BatchNorm-backward constants A,Bc,Cc from dL/dz (g) and the pre-BN tensor y (float64)."""
    mean = y_nchw.mean(dim=(0, 2, 3))
    var = y_nchw.var(dim=(0, 2, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    yhat = (y_nchw - mean[None, :, None, None]) * rstd[None, :, None, None]
    c1 = g_nchw.mean(dim=(0, 2, 3))
    c2 = (g_nchw * yhat).mean(dim=(0, 2, 3))
    A = gamma * rstd
    return torch.cat([A, -A * rstd * c2, A * (mean * rstd * c2 - c1)]), mean, rstd


@pytest.mark.parametrize("kind", ["dw", "pw"])
@pytest.mark.parametrize("B,H,W", [(3, 20, 76), (2, 7, 25), (1, 5, 3), (4, 17, 10)])
def test_conv_bwd_layers(nat, kind, B, H, W):
    """Two-layer chain  y_in -> bn_in -> relu -> conv -> bn -> (z) ; autograd supplies dL/dz_in,
    dW, dgamma_in, dbeta_in for an arbitrary dL/dz."""
    gen = torch.Generator().manual_seed(21)
    y_in = torch.randn(B, 64, H, W, generator=gen, dtype=torch.float64)
    bn_in = torch.nn.BatchNorm2d(64).double()
    bn_out = torch.nn.BatchNorm2d(64).double()
    with torch.no_grad():
        for bn in (bn_in, bn_out):
            bn.weight.copy_(torch.rand(64, generator=gen, dtype=torch.float64) + 0.5)
            bn.bias.copy_(torch.randn(64, generator=gen, dtype=torch.float64) * 0.3)
    if kind == "dw":
        w = (torch.randn(64, 1, 3, 3, generator=gen, dtype=torch.float64) * 0.3).requires_grad_(True)
    else:
        w = (torch.randn(64, 64, 1, 1, generator=gen, dtype=torch.float64) * 0.2).requires_grad_(True)
    z_in = bn_in(y_in)
    z_in.retain_grad()
    a = torch.relu(z_in)
    y = F.conv2d(a, w, padding=1, groups=64) if kind == "dw" else F.conv2d(a, w)
    z = bn_out(y)
    g = torch.randn(B, 64, H, W, generator=gen, dtype=torch.float64) * (torch.rand(B, 64, H, W, generator=gen) > 0.4)
    (z * g).sum().backward()

    coef, _, _ = _coef_from(g, y.detach(), bn_out.weight.detach())
    mean_in = y_in.mean(dim=(0, 2, 3))
    rstd_in = 1.0 / torch.sqrt(y_in.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    scale_in = bn_in.weight.detach() * rstd_in
    ss_in = torch.cat([scale_in, bn_in.bias.detach() - mean_in * scale_in])
    mr_in = torch.cat([mean_in, rstd_in])
    scratch = nat.layer_scratch(DEV)
    args = dict(y_out=cu(nhwc(y.detach())), coef=cu(coef), y_in=cu(nhwc(y_in)), ss_in=cu(ss_in), mr_in=cu(mr_in),
                gamma_in=cu(bn_in.weight.detach()), w=cu(w.detach()), scratch=scratch)
    if kind == "dw":
        g_in, dw, coef_in, dgamma, dbeta = nat.dwconv3x3_bwd(cu(nhwc(g)), **args)
    else:
        g_in, dw, coef_in, dgamma, dbeta = nat.pwconv1x1_bwd(cu(nhwc(g)), None, ss_out=None, **args)
    torch.cuda.synchronize()
    assert rel_err(g_in.cpu(), nhwc(z_in.grad)) < 2e-5, "dL/dz_in"
    assert rel_err(dw.cpu().reshape(-1), w.grad.reshape(-1)) < 2e-5, "dW"
    assert rel_err(dgamma.cpu(), bn_in.weight.grad) < 2e-5, "dgamma_in"
    assert rel_err(dbeta.cpu(), bn_in.bias.grad) < 2e-5, "dbeta_in"
    coef_ref, _, _ = _coef_from(z_in.grad, y_in, bn_in.weight.detach())
    assert np.abs(coef_in.cpu().numpy() - coef_ref.numpy()).max() < 2e-5 * (coef_ref.abs().max().item() + 1e-9) + 1e-9


# --------------------------------------------------------------------------- whole model
def _native_model_run(nat, sd, x, dlogits, dropout_p, seed, step, training=True):
    """Run ww_cnn_small_fwd/bwd with parameters from a state_dict (torch CPU tensors)."""
    names = ["stem.conv.weight", "stem.bn.weight", "stem.bn.bias", "stem.bn.running_mean", "stem.bn.running_var"]
    for i in range(4):
        names += [f"blocks.{i}.dw.weight"] + [f"blocks.{i}.dw_bn.{k}" for k in
                                              ("weight", "bias", "running_mean", "running_var")]
        names += [f"blocks.{i}.pw.weight"] + [f"blocks.{i}.pw_bn.{k}" for k in
                                              ("weight", "bias", "running_mean", "running_var")]
    names += ["classifier.weight", "classifier.bias"]
    assert len(names) == nat.CNN_SMALL_NPTR
    params = [sd[n].detach().float().to(DEV).contiguous() for n in names]
    grads = [torch.zeros_like(p) for p in params]
    xg = x.float().to(DEV).contiguous()
    B, Fd, T = x.shape[0], x.shape[2], x.shape[3]
    ws = torch.empty(nat.cnn_small_workspace_bytes(B, Fd, T) // 4, dtype=torch.float32, device=DEV)
    logits = torch.empty(B, 2, dtype=torch.float32, device=DEV)
    pa, ga = nat.ptr_array(params), nat.ptr_array(grads)
    nat.cnn_small_fwd(pa, xg, ws, logits, training=training, dropout_p=dropout_p, seed=seed, step=step)
    if training and dlogits is not None:
        nat.cnn_small_bwd(pa, ga, xg, dlogits.float().to(DEV).contiguous(), ws, dropout_p=dropout_p, seed=seed,
                          step=step)
    torch.cuda.synchronize()
    return logits.cpu(), dict(zip(names, params)), dict(zip(names, grads))


def _device_relu_masks(nat, model, x):
    """ReLU decisions the device takes for this model/input: run the conv stack layer by layer through the
    C-ABI and evaluate z = fma(y, scale, shift) > 0 exactly as the kernels do (fp32, single rounding)."""
    convs = [model.stem.conv] + [c for blk in model.blocks for c in (blk.dw, blk.pw)]
    bns = [model.stem.bn] + [c for blk in model.blocks for c in (blk.dw_bn, blk.pw_bn)]
    scratch = nat.layer_scratch(DEV)
    keep, masks = [], []
    y_prev = ss_prev = None
    for l, (conv, bn) in enumerate(zip(convs, bns)):
        t = [cu(bn.weight.detach()), cu(bn.bias.detach()), torch.zeros(64, device=DEV), torch.ones(64, device=DEV)]
        keep.append(t)
        h = nat.make_bn(*t)
        if l == 0:
            y, ss, _ = nat.conv_stem_fwd(cu(x), cu(conv.weight.detach()), h, scratch)
        else:
            fn = nat.dwconv3x3_fwd if l % 2 == 1 else nat.pwconv1x1_fwd
            y, ss, _ = fn(y_prev, ss_prev, cu(conv.weight.detach()), h, scratch)
        yc, sc = y.cpu().double(), ss.cpu().double()
        z32 = (yc * sc[:64] + sc[64:]).float()          # product+sum exact in float64 -> one rounding == fmaf
        masks.append((z32 > 0).permute(0, 3, 1, 2).contiguous())
        y_prev, ss_prev = y, ss
    return masks


def _oracle_with_masks(model, x, masks, keep_mask, p):
    """float64 forward of the oracle where ReLU is 'multiply by the given mask'."""
    convs = [model.stem.conv] + [c for blk in model.blocks for c in (blk.dw, blk.pw)]
    bns = [model.stem.bn] + [c for blk in model.blocks for c in (blk.dw_bn, blk.pw_bn)]
    a = x
    for conv, bn, m in zip(convs, bns, masks):
        a = bn(conv(a)) * m.double()
    pooled = a.mean(dim=(2, 3))
    if keep_mask is not None:
        pooled = pooled * torch.from_numpy(keep_mask.astype(np.float64) / (1.0 - float(np.float32(p))))
    return model.classifier(pooled)


@pytest.mark.parametrize("B,Fd,T,p", [(4, 40, 151, 0.0), (3, 40, 151, 0.3), (2, 13, 50, 0.3), (5, 64, 50, 0.0)])
def test_cnn_small_fwd_bwd_matches_oracle(nat, B, Fd, T, p):
    """ww_cnn_small_fwd/bwd vs float64 autograd of the oracle.

    An activation within fp32 round-off of 0 can land on either side of the ReLU (measured: 1 element in
    3.5e6 at (4,40,151), z = 1.1e-6); at these tiny batches BatchNorm-backward sums cancel so strongly that
    one such flip moves upstream gradients by ~1e-2.  The gradient comparison therefore hands the float64
    oracle the DEVICE's ReLU decisions; everything else (conv, BN statistics, BN backward, pooling, dropout,
    classifier) is compared as is.  The unmodified oracle is still checked on the logits."""
    from oracle.cnn_small import CNNSmallOracle, dropout_keep_mask
    torch.manual_seed(99)
    model = CNNSmallOracle(dropout=p, dropout_seed=77).double()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, 1, Fd, T, generator=gen, dtype=torch.float64) * 2 - 4
    dlog = torch.randn(B, 2, generator=gen, dtype=torch.float64) / B
    model.train()
    masks = _device_relu_masks(nat, model, x)
    keep = dropout_keep_mask(B, 64, p, 77, 4) if p > 0 else None
    model.load_state_dict(sd0)                       # undo the running-stat updates of nothing (bn untouched)
    out_m = _oracle_with_masks(model, x, masks, keep, p)
    out_m.backward(dlog)
    grads_ref = {n: prm.grad.clone() for n, prm in model.named_parameters()}
    bufs_ref = {n: b.clone() for n, b in model.named_buffers()}
    model.load_state_dict(sd0)
    model.dropout_step = 4
    out_plain = model(x).detach()

    logits, params, grads = _native_model_run(nat, sd0, x, dlog, p, seed=77, step=4)
    assert rel_err(logits, out_plain) < 5e-5, "logits vs unmodified oracle"
    assert rel_err(logits, out_m.detach()) < 5e-6, "logits vs oracle with device masks"
    worst = {n: rel_err(grads[n].cpu(), g) for n, g in grads_ref.items()}
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, f"gradient mismatch: {bad}"
    # running statistics updated like nn.BatchNorm2d
    for n, buf in bufs_ref.items():
        if n.endswith("running_mean") or n.endswith("running_var"):
            assert np.abs(params[n].cpu().numpy() - buf.numpy()).max() < 1e-4 * (buf.abs().max().item() + 1), n
    # eval-mode forward
    model.eval()
    sd1 = {k: v.clone() for k, v in model.state_dict().items()}
    logits_e, _, _ = _native_model_run(nat, sd1, x, None, p, seed=0, step=0, training=False)
    assert rel_err(logits_e, model(x).detach()) < 5e-5, "eval logits"
