"""GPU: fp16 activation-storage mode (WW_ACT_F16; BASELINE config 5 names fp16) -- the reference's own reduced precision is
fp16 autocast + ``GradScaler`` (src/training/trainer.py:172,182-193; src/training/optimizer_factory.py:403-420).  Here the
16-bit type is the STORAGE / matrix-operand type of the HIP kernels (arithmetic, statistics, parameters stay fp32) and the
scaler lives on the device: the loss kernel multiplies dL/dlogits by the scale, the fused optimizer divides it out, skips
on overflow and applies GradScaler's growth / backoff rule.  Tolerances are fp16-sized (11-bit mantissa) and stated."""
import numpy as np
import pytest
import torch

from tests.golden_util import load_trace, make_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos(a, b):
    return (a @ b / (a.norm() * b.norm())).item()


@pytest.mark.parametrize("B,Fd,T,p", [(4, 40, 151, 0.3), (6, 13, 50, 0.0), (5, 8, 12, 0.0)])
def test_cnn_small_fp16_close_to_oracle(B, Fd, T, p):
    """Whole model, fp16 storage vs the float64 oracle: logits within 4e-3 of their scale, gradient direction cos > 0.9995
    (bf16 storage holds 3e-2 / 0.995: three more mantissa bits).  The upstream gradient is multiplied by 1024 on the way in
    and divided out of the results -- what the loss scale does in training -- so the stored gradients sit in fp16's range."""
    from oracle.cnn_small import CNNSmallOracle
    from wakeword_trainer_home_amd.models import create_model
    torch.manual_seed(5)
    oracle = CNNSmallOracle(dropout=p, dropout_seed=3).double()
    model = create_model("cnn_small", dropout=p, dropout_seed=3, act_dtype="fp16")
    model.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()})
    model.to(DEV).train()
    oracle.train()
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(B, 1, Fd, T, generator=gen, dtype=torch.float64) * 2 - 4
    dlog = torch.randn(B, 2, generator=gen, dtype=torch.float64) / B
    out = model(x.float().to(DEV))
    out.backward((dlog * 1024.0).float().to(DEV))
    ref = oracle(x)
    ref.backward(dlog)
    assert (out.detach().cpu().double() - ref.detach()).abs().max() < 4e-3 * max(ref.abs().max().item(), 1.0)
    gn = torch.cat([q.grad.flatten().cpu().double() for q in model.parameters()]) / 1024.0
    go = torch.cat([q.grad.flatten() for q in oracle.parameters()])
    assert torch.isfinite(gn).all()
    assert _cos(gn, go) > (0.9995 if Fd * T > 1000 else 0.998), _cos(gn, go)      # (13 x 50 maps: 0.99949-0.99952 measured)
    assert abs(gn.norm().item() / go.norm().item() - 1.0) < 1e-2


def _trainer(tmp_path, golden_dir, amp, init_scale=None):
    from wakeword_trainer_home_amd.config import WakewordConfig
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    meta, tr = load_trace(golden_dir, "default_b128")
    cfg = WakewordConfig()
    for sec in ("loss", "optimizer", "training"):
        for k, v in meta["cfg"][sec].items():
            if hasattr(getattr(cfg, sec), k):
                setattr(getattr(cfg, sec), k, v)
    cfg.model.architecture, cfg.optimizer.mixed_precision, cfg.optimizer.amp_dtype = "cnn_small", True, amp
    model = create_model("cnn_small", dropout=0.0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tr["init"].items()})
    xtr, ytr = make_inputs(meta["train_seed"], meta["n_train"])
    xva, yva = make_inputs(meta["val_seed"], meta["n_val"])
    DL, TD = torch.utils.data.DataLoader, torch.utils.data.TensorDataset
    t = Trainer(model, DL(TD(xtr, ytr), batch_size=128), DL(TD(xva, yva), batch_size=128), cfg, checkpoint_dir=tmp_path, device=DEV)
    if init_scale is not None:
        from wakeword_trainer_home_amd import _native as nat
        t.scaler.state.copy_(nat.loss_scale_new("cpu", init_scale))
    return t, meta, tr


def test_trainer_fp16_tracks_reference_trace(golden_dir, tmp_path):
    """mixed_precision=True + amp_dtype='fp16' through the Trainer on the reference's B=128 trace (captured from the real
    reference Trainer in fp32): per-step loss within 2.5e-4 (10x the 2.2e-5 measured on MI355X); the scaler is the device one, its scale stayed at GradScaler's
    initial 65536 and its growth tracker counted every applied step; the checkpoint's scaler_state_dict has GradScaler's keys."""
    from wakeword_trainer_home_amd import _native as nat
    from wakeword_trainer_home_amd.training.optimizer_factory import DeviceGradScaler
    t, meta, tr = _trainer(tmp_path, golden_dir, "fp16")
    assert t.model.act == nat.ACT_F16 and isinstance(t.scaler, DeviceGradScaler)
    losses = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: losses.append(l)})())
    t.train()
    d = np.abs(np.array(losses) - tr["step_loss"])
    assert len(losses) == len(tr["step_loss"]) and d.max() < 2.5e-4, d
    sd = t.scaler.state_dict()
    assert set(sd) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    assert sd["scale"] == 65536.0 and sd["_growth_tracker"] == len(losses) == t.optimizer.step_count()
    ck = torch.load(tmp_path / "best_model.pt", map_location="cpu", weights_only=False)
    assert ck["scaler_state_dict"]["scale"] == 65536.0
    print(f"fp16 storage: max |loss - ref| = {d.max():.2e}")


def test_overflowing_scale_backs_off_and_skips_like_gradscaler(golden_dir, tmp_path):
    """Start from a loss scale far too large (2^40): the scaled gradients overflow fp16 -> the step is skipped on the device
    (parameters bit-unchanged, optimizer step count unchanged) and the scale halves, again and again, until a step fits --
    GradScaler.update()'s backoff.  Then training proceeds; growth_interval applied steps later the scale doubles."""
    from wakeword_trainer_home_amd import _native as nat
    t, meta, tr = _trainer(tmp_path, golden_dir, "fp16", init_scale=2.0 ** 40)
    t.scaler.state.copy_(nat.loss_scale_new("cpu", 2.0 ** 40, growth_interval=3))
    x, y = make_inputs(meta["train_seed"], 128)
    t.model.train()
    before = t.model.flat_param.clone()
    scales, applied = [], []
    for i in range(40):
        t._step_native(x, y, i)
        t._flush_pending()
        scales.append(t.scaler.get_scale())
        applied.append(t.optimizer.step_count())
        if applied[-1] == 0:
            assert torch.equal(t.model.flat_param, before)        # skipped steps leave the parameters alone
    first = next(i for i, a in enumerate(applied) if a > 0)
    assert first >= 5                                             # 2^40 has to come down a long way (9 halvings measured)
    assert all(scales[i] == 2.0 ** (39 - i) for i in range(first))            # halved once per skipped step
    assert torch.isfinite(t.model.flat_param).all() and not torch.equal(t.model.flat_param, before)
    # growth: every 3 consecutive applied steps double the scale (an overflow in between resets the count and halves it)
    assert any(b > a for a, b in zip(scales[first:], scales[first + 1:])), scales
    assert applied[-1] >= 20


def test_fp16_graph_replay_is_bit_exact(tmp_path):
    """The loss scale and its slot are device state too: a replayed fp16 step equals the eager one bit for bit, scaler included."""
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    x, y = make_inputs(4, 16 * 6)
    batches = [(x[16 * i:16 * i + 16], y[16 * i:16 * i + 16]) for i in range(6)]
    out = []
    for graph in (False, True):
        cfg = get_preset("cnn_small_logmel40")
        cfg.training.epochs, cfg.training.batch_size, cfg.optimizer.warmup_epochs = 2, 16, 0
        cfg.optimizer.mixed_precision, cfg.optimizer.amp_dtype, cfg.training.hip_graph = True, "fp16", graph
        cfg.training.hip_graph_auto = False
        torch.manual_seed(2)
        model = create_model("cnn_small", dropout=0.3, dropout_seed=1)
        t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path / str(graph), device=DEV)
        losses = []
        t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: losses.append(l)})())
        t.train()
        out.append((losses, model.flat_param.clone(), t.scaler.state_dict(), t._graph is not None))
    assert out[1][3] and not out[0][3]
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
