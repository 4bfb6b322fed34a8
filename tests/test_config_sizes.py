"""BASELINE.json's configurations at the sizes they name, against the CPU oracles:

  config 2  cnn_small, bf16 storage, batch 512 -- ONE training step of exactly what bench.py times, vs the float64 oracle step
  config 3  MobileNetV3 at its per-GPU batch 256 (2048 / 8 ranks), training step vs oracle/mobilenetv3.py
  config 4  Large-Dataset preset's input stage (src/config/presets.py:94-151: batch 128, 2.5 s clips, noise p=0.4 at
            10-20 dB, RIR p=0.25): on-GPU RIR + background mix vs the float64 oracle on a clip subset, bit-exact choices for all
  config 5  CRNN at its per-GPU batch 512 (4096 / 8), fp32 / bf16 / fp16 storage, training step vs oracle/crnn.py

Train-mode BatchNorm couples a batch, so the oracles run the WHOLE batch (seconds on the box's host cores); bounds are
stated per test and are whole-vector / per-tensor, as the small-size tests' are."""
import numpy as np
import pytest
import torch

from tests.golden_util import make_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos(a, b):
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def test_config2_bf16_training_step_at_batch_512_matches_oracle():
    """bench.py's configuration: bf16 activation/gradient storage, B=512 feature maps (512,1,40,151), dropout 0.3 on.
    This is the shape that selects k_pw_bwd_bf16's WIDE_IMG variant, its y_out recompute and the pooled-gradient shortcut
    of the last layer.  Reference step: src/training/trainer.py:165-203.  Bounds = 10x what an MI355X measures (r03: loss 7e-6,
    gradient norm 1.5e-5 relative, cos 0.99997; the line this test prints is kept under profiles/): loss within 7e-5, gradient
    norm within 2e-4, direction of the whole gradient cos > 0.9997, of every conv /
    classifier weight tensor > 0.95 (worst: the stem's, 0.969 -- the last stop of the backward chain, and its input has a
    mean of -4 against a spread of 2: sum(dy) is exactly 0 behind a BatchNorm, so mean * sum(rounding errors of dy) is pure
    noise on top of sum(dy * (x - mean))) and of every BatchNorm weight / bias gradient > 0.9 (those 64-vectors are sums with
    structural cancellation -- the consumer's BatchNorm backward makes its input gradient sum to zero per channel, so for a
    1x1 consumer sum(dL/da) = 0 exactly and dbeta = sum over the z > 0 pixels only -- hence they carry the bf16 rounding
    noise of ~780 k addends against a small total; measured 0.965 at worst), BatchNorm running statistics within 1 % of
    their scale, parameters after the SGD update within 3e-4."""
    from wakeword_trainer_home_amd import _native as nat
    from wakeword_trainer_home_amd.models import create_model, create_loss_function
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss
    B = 512
    torch.manual_seed(4)
    model = create_model("cnn_small", dropout=0.3, dropout_seed=9, act_dtype="bf16").to(DEV)
    oracle = CNNSmallOracle(dropout=0.3, dropout_seed=9).double()
    oracle.load_state_dict({k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in model.state_dict().items()})
    x, y = make_inputs(21, B)
    model.train()
    oracle.train()
    opt = create_optimizer(model, "sgd", learning_rate=0.05, weight_decay=1e-2, momentum=0.9)
    crit = create_loss_function("cross_entropy", label_smoothing=0.05, device=DEV)
    opt.zero_grad(set_to_none=True)
    stats = model.train_step_native(x.to(DEV), y.to(DEV), crit)
    g_dev = {n: p.grad.detach().cpu().double().clone() for n, p in model.named_parameters()}    # before the in-place clip
    opt.step(max_norm=1.0, stats=stats)
    s = nat.decode_stats(stats.cpu())

    oopt = torch.optim.SGD(oracle.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2, nesterov=True)
    oopt.zero_grad(set_to_none=True)
    out = oracle(x.double())
    loss = TorchLoss("cross_entropy", eps=0.05)(out, y)
    loss.backward()
    g_ref = {n: p.grad.detach().clone() for n, p in oracle.named_parameters()}
    gn_ref = float(torch.nn.utils.clip_grad_norm_(oracle.parameters(), 1.0))
    oopt.step()
    acc_ref = float((out.argmax(1) == y).float().mean())

    assert s["found_inf"] == 0.0 and s["count"] == B
    assert abs(s["loss"] - loss.item()) <= 7e-5, (s["loss"], loss.item())
    assert abs(s["grad_norm"] - gn_ref) <= 2e-4 * gn_ref, (s["grad_norm"], gn_ref)
    assert abs(s["correct"] / B - acc_ref) <= 0.02                     # logits within a bf16 step of 0 may flip the argmax
    gd, go = torch.cat([g_dev[n].flatten() for n in g_ref]), torch.cat([g.flatten() for g in g_ref.values()])
    assert _cos(gd, go) > 0.9997, _cos(gd, go)
    per = {n: _cos(g_dev[n].flatten(), g.flatten()) for n, g in g_ref.items() if g.norm() > 1e-6 * go.norm()}
    worst_w = min((c, n) for n, c in per.items() if "bn" not in n)
    worst_bn = min((c, n) for n, c in per.items() if "bn" in n)
    assert worst_w[0] > 0.95, worst_w
    assert worst_bn[0] > 0.93, worst_bn
    worst = (worst_w, worst_bn)
    for (n, b), (_, c) in zip(model.named_buffers(), oracle.named_buffers()):
        if b.is_floating_point():
            assert (b.cpu().double() - c).abs().max().item() <= 1e-2 * max(c.abs().max().item(), 1e-3), n
    for (n, p), q in zip(model.named_parameters(), oracle.parameters()):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() <= 3e-4, n
    print(f"config 2 bf16 B=512: loss {s['loss']:.6f} vs {loss.item():.6f}, |g| {s['grad_norm']:.5f} vs {gn_ref:.5f}, "
          f"cos {_cos(gd, go):.5f}, worst tensor {worst}")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_config3_mobilenetv3_training_step_at_per_gpu_batch_256(mode):
    """config 3: global batch 2048 over 8 ranks = 256 per GPU.  Bounds <= 10x the measured values (r03, fp32: logits 7e-7, loss
    equal to 6 digits, gradient 1.6e-6 of its norm, running statistics 1.1e-7; bf16 matrix mode -- operands rounded, fp32
    accumulation --: logits 1.4e-2, loss 2.2e-5, cos 0.9987, running statistics 2.5e-3 of a standard deviation)."""
    from wakeword_trainer_home_amd.models import create_model
    from oracle.mobilenetv3 import MobileNetV3Oracle
    B = 256
    torch.manual_seed(11)
    model = create_model("mobilenetv3", dropout=0.3, dropout_seed=2, mode=mode).to(DEV)
    oracle = MobileNetV3Oracle(dropout=0.3, seed=2)
    oracle.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in model.state_dict().items()})
    x, y = make_inputs(8, B)
    model.train()
    oracle.train()
    out = model(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
    loss.backward()
    ref = oracle(x, step=0, training=True)
    lo = torch.nn.functional.cross_entropy(ref, y)
    lo.backward()
    tol, ltol = (7e-6, 5e-6) if mode == "fp32" else (5e-2, 2e-4)
    derr = (out.detach().cpu().double() - ref.detach()).abs().max().item()
    assert derr <= tol * max(ref.abs().max().item(), 1.0), derr
    assert abs(loss.item() - lo.item()) <= ltol, (loss.item(), lo.item())
    gd = torch.cat([p.grad.flatten().cpu().double() for p in model.parameters()])
    go = torch.cat([p.grad.flatten() for p in oracle.parameters()])
    rel = ((gd - go).norm() / go.norm()).item()
    if mode == "fp32":
        assert rel <= 2e-5, rel
    else:
        assert _cos(gd, go) > 0.99, _cos(gd, go)
    # running statistics after the step (torchvision's momentum 0.01, initial mean 0 / var 1), measured in units of the
    # channel's batch standard deviation -- the scale a consumer of the normalised activation sees (a channel mean can be
    # ~0 next to unit-sized values, so a ratio of means says nothing).  bf16 matrix mode: the rounding of a weight is common
    # to every pixel and does not average out of a channel mean: 2e-2 of a standard deviation
    worst_rs = 0.0
    ob = dict(oracle.named_buffers())
    for n, b in model.named_buffers():
        if not n.endswith("running_mean"):
            continue
        var_b = ((ob[n.replace("running_mean", "running_var")].double() - 0.99) / 0.01).clamp_min(1e-12)
        r = ((b.cpu().double() - ob[n].double()).abs() / 0.01 / var_b.sqrt()).max().item()
        worst_rs = max(worst_rs, r)
        assert r <= (2e-6 if mode == "fp32" else 2e-2), (n, r)
        # (the batch variance is recovered from fp32 running_var ~ 1: one ulp of it is 6e-6 of variance -- percent-sized next
        # to the 2e-4 variance of a nearly dead channel; the denominator is floored at 0.05 so those do not set the bound)
        rv = (model.get_buffer(n.replace("running_mean", "running_var")).cpu().double() - ob[n.replace("running_mean", "running_var")].double()).abs() / 0.01 / (var_b + 0.05)
        assert rv.max().item() <= (1e-3 if mode == "fp32" else 5e-2), (n, rv.max().item())
    print(f"config 3 mobilenetv3 B=256 {mode}: logits err {derr:.2e}, loss {loss.item():.6f} vs {lo.item():.6f}, grad rel {rel:.2e}, "
          f"cos {_cos(gd, go):.6f}, running stats worst rel {worst_rs:.2e}")


def test_config4_large_dataset_augmentation_at_batch_128():
    """Large-Dataset preset input stage: 128 clips x 2.5 s (40 000 samples), RIR p=0.25 (0.25 s = 4000 taps), background
    noise p=0.4 at 10-20 dB.  Philox choices (which RIR / noise clip / offset / SNR) bit-exact for ALL 128 clips; the
    signal against the float64 oracle on a subset that contains every combination (plain, RIR only, noise only, both),
    1e-4 absolute; untouched clips are bit-exact copies."""
    from wakeword_trainer_home_amd import _native as nat
    from oracle import audio_augment as oa
    B, N, L = 128, 40000, 4000
    rng = np.random.default_rng(44)
    x = (0.2 * rng.standard_normal((B, N))).astype(np.float32)
    rirs = (rng.standard_normal((16, L)) * np.exp(-np.arange(L) / 700.0)).astype(np.float32)
    noises = (0.1 * rng.standard_normal((8, 10 * 16000))).astype(np.float32)
    kw = dict(rir_prob=0.25, noise_prob=0.4, snr_min_db=10.0, snr_max_db=20.0, seed=2024, step=31, sample_offset=3 * B)
    t = lambda a: torch.from_numpy(a).to(DEV)
    rd = t(rirs)
    out, ch = nat.audio_augment(t(x), rd, t(noises), want_choices=True, rir_spectra=nat.audio_rir_spectra(rd), **kw)
    out, ch = out.cpu().numpy(), ch.cpu().numpy()
    # choices of every clip from the oracle's integer law (no convolution needed for them)
    ref_ch = oa.audio_choices(B, N, len(rirs), len(noises), noises.shape[1], 0.25, 0.4, 10.0, 20.0, 2024, 31, 3 * B)
    assert np.array_equal(ch[:, 0], ref_ch["rir"]) and np.array_equal(ch[:, 1], ref_ch["noise"])
    assert np.array_equal(ch[:, 2], ref_ch["offset"])
    assert np.allclose(ch[:, 3].copy().view(np.float32), ref_ch["snr_db"], rtol=0, atol=2e-6)
    kinds = {}
    for i in range(B):
        kinds.setdefault((ch[i, 0] >= 0, ch[i, 1] >= 0), []).append(i)
    assert len(kinds) == 4, "the draw should contain all four combinations at B=128"
    sub = sorted(i for v in kinds.values() for i in v[:3])
    for i in sub:
        ref, _ = oa.audio_augment(x[i:i + 1], rirs, noises, 0.25, 0.4, 10.0, 20.0, 2024, 31, 3 * B + i)
        err = np.abs(out[i] - ref[0]).max()
        assert err <= 1e-4, (i, err)
    plain = np.array(kinds[(False, False)])
    assert np.array_equal(out[plain], np.clip(x[plain], -1, 1))
    assert np.isfinite(out).all() and np.abs(out).max() <= 1.0


@pytest.mark.parametrize("act", ["fp32", "bf16", "fp16"])
def test_config5_crnn_training_step_at_per_gpu_batch_512(act):
    """config 5: global batch 4096 over 8 ranks = 512 per GPU; conv front-end + bidirectional 2-layer GRU, dropout on.
    Bounds <= 10x the measured values (r03: fp32 logits 2.3e-7, loss equal to 6 digits, recurrent gradients 2.6e-7 of their
    norm; bf16 storage + bf16 matrix operands: logits 1.5e-3, loss 3e-5, recurrent gradients 2.5e-3, conv-stack direction cos
    0.9949; fp16 storage + fp16 matrix operands (config 5 as BASELINE words it): logits 2.4e-4, loss 1e-6, recurrent 3.2e-4,
    conv cos 0.9994) -- the fp16 backward runs on the loss times 65536 (GradScaler's initial scale, what the Trainer's device
    scaler applies) and the gradients are divided by it.  Conv-stack gradient, fp32 mode: 5.5e-4 of its norm measured (single
    ReLU decisions of activations within round-off of zero move it, see test_gru.py), bound 5e-3."""
    from wakeword_trainer_home_amd.models import create_model
    from oracle.crnn import CRNNOracle
    B = 512
    torch.manual_seed(3)
    model = create_model("crnn", dropout=0.3, dropout_seed=4, act_dtype=act).to(DEV)
    oracle = CRNNOracle(dropout=0.3, seed=4)
    oracle.load_device_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    x, y = make_inputs(5, B)
    model.train()
    oracle.train()
    out = model(x.to(DEV))
    loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
    S = 65536.0 if act == "fp16" else 1.0
    (loss * S).backward()
    ref = oracle(x, step=0, training=True)
    lo = torch.nn.functional.cross_entropy(ref, y)
    lo.backward()
    tol, ltol, rtol = {"fp32": (2.5e-6, 5e-6, 3e-6), "bf16": (1.5e-2, 3e-4, 2.5e-2), "fp16": (2.5e-3, 2e-5, 3.2e-3)}[act]
    derr = (out.detach().cpu().double() - ref.detach()).abs().max().item()
    assert derr <= tol, derr
    assert abs(loss.item() - lo.item()) <= ltol, (loss.item(), lo.item())
    gd = torch.cat([p.grad.flatten().cpu().double() for n, p in model.named_parameters() if n.startswith("rnn.")]) / S
    go = torch.cat([p.grad.flatten() for p in oracle.rnn.parameters()])
    fd = torch.cat([p.grad.flatten().cpu().double() for n, p in model.named_parameters() if n.startswith("front.")]) / S
    fo = torch.cat([p.grad.flatten() for n, p in oracle.front.named_parameters() if not n.startswith("classifier")])
    assert torch.isfinite(gd).all() and torch.isfinite(fd).all()
    rrel, frel = ((gd - go).norm() / go.norm()).item(), ((fd - fo).norm() / fo.norm()).item()
    assert rrel <= rtol, rrel
    if act == "fp32":
        assert frel <= 5e-3, frel
    elif act == "bf16":
        assert _cos(fd, fo) > 0.97, _cos(fd, fo)
    else:
        assert _cos(fd, fo) > 0.995, _cos(fd, fo)
    print(f"config 5 crnn B=512 {act}: logits err {derr:.2e}, loss {loss.item():.6f} vs {lo.item():.6f}, "
          f"rnn grad rel {rrel:.2e} cos {_cos(gd, go):.5f}, conv grad rel {frel:.2e} cos {_cos(fd, fo):.5f}")
