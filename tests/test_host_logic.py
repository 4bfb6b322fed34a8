"""CPU: the host-side mirror of the reference interface (Trainer, factories, metrics, schedulers,
checkpoints) against the fixtures captured from the real reference classes, plus the C-ABI export check.
The device work is not exercised here (no GPU): the Trainer drives the plain-torch oracle model through
its generic-module step, which must reproduce the reference Trainer's trace exactly."""
import ctypes
import json
import os
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.golden_util import load_trace, make_inputs

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture()
def no_gpu_gate(monkeypatch):
    import wakeword_trainer_home_amd.training.trainer as T
    monkeypatch.setattr(T, "enforce_cuda", lambda: None)
    return T


def _cfg_from_meta(meta):
    from wakeword_trainer_home_amd.config import WakewordConfig
    cfg = WakewordConfig()
    for sec in ("loss", "optimizer", "training"):
        for k, v in meta["cfg"][sec].items():
            if hasattr(getattr(cfg, sec), k):
                setattr(getattr(cfg, sec), k, v)
    cfg.model.architecture = "cnn_small"
    return cfg


class _Rec:
    def __init__(self):
        self.loss, self.acc, self.epochs, self.starts = [], [], [], []

    def on_epoch_start(self, epoch):
        self.starts.append(epoch)

    def on_batch_end(self, batch_idx, loss, acc):
        self.loss.append(loss)
        self.acc.append(acc)

    def on_epoch_end(self, epoch, train_loss, val_loss, val_metrics):
        self.epochs.append(dict(epoch=epoch, train_loss=train_loss, val_loss=val_loss,
                                val_metrics=val_metrics.to_dict()))


@pytest.mark.parametrize("tag", ["default_b16", "focal_b16", "sgd_b8"])
def test_trainer_reproduces_reference_trace_on_cpu(golden_dir, tmp_path, no_gpu_gate, tag):
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss
    meta, tr = load_trace(golden_dir, tag)
    cfg = _cfg_from_meta(meta)
    model = CNNSmallOracle(dropout=0.0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tr["init"].items()})
    xtr, ytr = make_inputs(meta["train_seed"], meta["n_train"])
    xva, yva = make_inputs(meta["val_seed"], meta["n_val"])
    DL, TD = torch.utils.data.DataLoader, torch.utils.data.TensorDataset
    crit = TorchLoss(cfg.loss.loss_function, eps=cfg.loss.label_smoothing, alpha=cfg.loss.focal_alpha,
                     gamma=cfg.loss.focal_gamma)
    t = no_gpu_gate.Trainer(model, DL(TD(xtr, ytr), batch_size=meta["batch"]), DL(TD(xva, yva), batch_size=meta["batch"]),
                            cfg, checkpoint_dir=tmp_path, device="cpu", criterion=crit)
    rec = _Rec()
    t.add_callback(rec)
    res = t.train()
    assert type(t.optimizer).__name__ == meta["optimizer"]
    assert (type(t.scheduler).__name__ if t.scheduler else None) == meta["scheduler"]
    np.testing.assert_allclose(rec.loss, tr["step_loss"], atol=1e-6)
    np.testing.assert_allclose(rec.acc, tr["step_acc"], atol=1e-7)
    assert t.state.global_step == meta["global_step"]
    for k, ref in meta["history"].items():
        np.testing.assert_allclose(res["history"][k], ref, atol=2e-6, err_msg=k)
    for k in ("final_epoch", "best_f1_epoch", "best_fpr_epoch"):
        assert res[k] == meta[k], k
    for k in ("best_val_loss", "best_val_f1", "best_val_fpr"):
        assert res[k] == pytest.approx(meta[k], abs=2e-6), k
    assert set(res) == {"history", "final_epoch", "best_val_loss", "best_val_f1", "best_val_fpr", "training_time",
                        "best_f1_epoch", "best_fpr_epoch"}
    assert [e["epoch"] for e in rec.epochs] == [e["epoch"] for e in meta["epochs_rec"]]
    for got, ref in zip(rec.epochs, meta["epochs_rec"]):
        assert got["val_metrics"].keys() == ref["val_metrics"].keys()
        for k, v in ref["val_metrics"].items():
            assert got["val_metrics"][k] == pytest.approx(v, abs=1e-9), k
    # checkpoints: same files, same key set, same TrainingState fields (G5)
    assert sorted(p.name for p in tmp_path.iterdir()) == meta["files"]
    ck = torch.load(tmp_path / "best_model.pt", map_location="cpu", weights_only=False)
    # the reference's key set plus the build's one documented extension (the Philox launched-step counter)
    assert sorted(set(ck.keys()) - {"launched_steps"}) == meta["ckpt_keys"] and "launched_steps" in ck
    assert sorted(vars(ck["state"]).keys()) == meta["state_fields"]
    # final parameters equal the reference run's.  Adam divides by sqrt(v): on weights whose gradient is at
    # round-off level, a 1e-9 difference in the gradient (oracle focal restatement vs the reference's
    # formulation) becomes an lr-sized update, hence 2e-4 here although every loss agrees to 1e-6.
    for k, v in tr["final"].items():
        np.testing.assert_allclose(model.state_dict()[k].numpy(), v, atol=2e-4, err_msg=k)


def test_resume_continues_like_an_uninterrupted_run(golden_dir, tmp_path, no_gpu_gate):
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss
    meta, tr = load_trace(golden_dir, "default_b16")
    xtr, ytr = make_inputs(meta["train_seed"], meta["n_train"])
    xva, yva = make_inputs(meta["val_seed"], meta["n_val"])
    DL, TD = torch.utils.data.DataLoader, torch.utils.data.TensorDataset

    def mk(epochs_now, d):
        cfg = _cfg_from_meta(meta)
        model = CNNSmallOracle(dropout=0.0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in tr["init"].items()})
        t = no_gpu_gate.Trainer(model, DL(TD(xtr, ytr), batch_size=16), DL(TD(xva, yva), batch_size=16), cfg,
                                checkpoint_dir=d, device="cpu", criterion=TorchLoss("cross_entropy", eps=0.05))
        return t, cfg
    t1, cfg1 = mk(4, tmp_path / "a")
    cfg1.training.epochs = 2                      # stop after 2 of 4 epochs (scheduler built for 4)
    t1.train()
    t2, _ = mk(4, tmp_path / "b")
    rec = _Rec()
    t2.add_callback(rec)
    res = t2.train(resume_from=tmp_path / "a" / "checkpoint_epoch_002.pt")
    assert rec.starts == [2, 3] and res["final_epoch"] == 3
    np.testing.assert_allclose(rec.loss, tr["step_loss"][12:], atol=2e-6)
    with pytest.raises(FileNotFoundError):
        t2.load_checkpoint(tmp_path / "missing.pt")
    bad = tmp_path / "bad.pt"
    torch.save({"epoch": 1}, bad)
    with pytest.raises(ValueError, match="missing required keys"):
        t2.load_checkpoint(bad)


def test_batch_contract_and_error_policy(tmp_path, no_gpu_gate):
    """2-/3-tuples accepted, malformed / empty batches skipped, OOM skipped, other RuntimeError re-raised,
    other exceptions logged + skipped, non-finite loss skipped but counted in the denominator (Q5)."""
    from wakeword_trainer_home_amd.config import WakewordConfig
    cfg = WakewordConfig()
    cfg.training.epochs, cfg.optimizer.warmup_epochs = 1, 0
    lin = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 2))
    x, y = torch.randn(4, 1, 3, 4), torch.tensor([0, 1, 0, 1])
    batches = [(x, y), (x, y, [{"path": "a"}] * 4), "junk", (x[:0], y[:0]), (x * float("nan"), y), [x, y]]
    t = no_gpu_gate.Trainer(lin, batches, [(x, y)], cfg, checkpoint_dir=tmp_path, device="cpu",
                            criterion=torch.nn.CrossEntropyLoss())
    rec = _Rec()
    t.add_callback(rec)
    avg, acc = t.train_epoch(0)
    assert len(rec.loss) == 3 and t.state.global_step == 3
    assert avg == pytest.approx(sum(rec.loss) / 6)

    class Boom(torch.nn.Module):
        def __init__(self, exc):
            super().__init__()
            self.l, self.exc, self.n = torch.nn.Linear(12, 2), exc, 0

        def forward(self, v):
            self.n += 1
            if self.n == 2:
                raise self.exc
            return self.l(v.flatten(1))
    for exc, reraised in ((RuntimeError("HIP out of memory"), False), (KeyError("x"), False),
                          (RuntimeError("device-side assert"), True)):
        t = no_gpu_gate.Trainer(Boom(exc), [(x, y)] * 3, [(x, y)], cfg, checkpoint_dir=tmp_path, device="cpu",
                                criterion=torch.nn.CrossEntropyLoss())
        if reraised:
            with pytest.raises(RuntimeError):
                t.train_epoch(0)
        else:
            t.train_epoch(0)
            assert t.state.global_step == 2
    assert t.validate_epoch(0)[1].total_samples == 4
    t.val_loader = []
    loss, m = t.validate_epoch(0)
    assert loss == 0.0 and m.total_samples == 0


def test_scheduler_quirks_match_reference(golden_dir, no_gpu_gate):
    from wakeword_trainer_home_amd.config import WakewordConfig
    from wakeword_trainer_home_amd.training import optimizer_factory as of
    ref = json.loads((golden_dir / "g4_sched.json").read_text())
    val_losses = ref.pop("val_losses")
    for name, case in ref.items():
        cfg = WakewordConfig()
        cfg.training.epochs = len(val_losses)
        for k, v in case["cfg"].items():
            setattr(cfg.optimizer, k, v)
        opt, sch = of.create_optimizer_and_scheduler(torch.nn.Linear(4, 2), cfg)
        holder = type("H", (), {"scheduler": sch})()
        lrs = [of.get_learning_rate(opt)]
        for vl in val_losses:
            opt.step()
            no_gpu_gate.Trainer._update_scheduler(holder, vl)
            lrs.append(of.get_learning_rate(opt))
        np.testing.assert_allclose(lrs, case["lrs"], rtol=1e-12, err_msg=name)


def test_factory_validation_messages():
    from wakeword_trainer_home_amd.training import optimizer_factory as of
    m = torch.nn.Linear(2, 2)
    for kw, msg in ((dict(learning_rate=0), "Learning rate must be positive"), (dict(weight_decay=-1), "Weight decay"),
                    (dict(momentum=2), "Momentum"), (dict(betas=(0.9, 1.5)), "Betas"),
                    (dict(optimizer_name="lion"), "Unknown optimizer")):
        with pytest.raises(ValueError, match=msg):
            of.create_optimizer(m, **kw)
    opt = of.create_optimizer(m, "sgd")
    assert opt.defaults["nesterov"] is True
    for kw, msg in ((dict(epochs=0), "Epochs"), (dict(warmup_epochs=50), "Warmup epochs"), (dict(gamma=0), "Gamma"),
                    (dict(factor=1.0), "Factor"), (dict(scheduler_name="poly"), "Unknown scheduler")):
        with pytest.raises(ValueError, match=msg):
            of.create_scheduler(opt, **kw)
    assert of.create_scheduler(opt, "none") is None
    with pytest.raises(ValueError):
        of.clip_gradients(m, 0.0)


def test_metrics_match_reference(golden_dir):
    from wakeword_trainer_home_amd.training.metrics import MetricsCalculator, MetricsTracker, MetricMonitor, MetricResults
    cases = json.loads((golden_dir / "g3_metrics.json").read_text())
    for name, c in cases.items():
        z, y = torch.tensor(c["logits"]), torch.tensor(c["targets"])
        assert MetricsCalculator("cpu").calculate(z, y).to_dict() == pytest.approx(c["result"]), name
        tr = MetricsTracker("cpu")
        half = len(y) // 2
        tr.update(z[:half], y[:half]) if half else None
        tr.update(z[half:], y[half:])
        assert tr.compute().to_dict() == pytest.approx(c["result"]), name
    tr = MetricsTracker("cpu")
    assert tr.compute().total_samples == 0
    assert tr.get_best_epoch("f1_score") == (0, None)
    for f1, fpr in ((0.2, 0.5), (0.7, 0.3), (0.6, 0.1)):
        m = MetricResults.empty()
        m.f1_score, m.fpr = f1, fpr
        tr.save_epoch_metrics(m)
    assert tr.get_best_epoch("f1_score")[0] == 1 and tr.get_best_epoch("fpr")[0] == 2
    mon = MetricMonitor(window_size=3)
    for i in range(5):
        mon.update_batch(float(i), 1.0)
    assert mon.get_running_averages()["loss"] == pytest.approx(3.0)


def test_config_roundtrip_and_presets(tmp_path):
    from wakeword_trainer_home_amd.config import WakewordConfig, get_preset, list_presets
    c = get_preset("cnn_small_logmel40")
    assert c.model.architecture == "cnn_small" and c.data.n_mels == 40 and c.training.batch_size == 512
    c.save(tmp_path / "c.yaml")
    assert WakewordConfig.load(tmp_path / "c.yaml").to_dict() == c.to_dict()
    d = WakewordConfig()
    assert (d.data.n_fft, d.data.hop_length, d.data.n_mels, d.optimizer.optimizer, d.optimizer.warmup_epochs,
            d.optimizer.min_lr, d.loss.label_smoothing, d.optimizer.gradient_clip) == \
           (1024, 160, 128, "adamw", 3, 3e-4, 0.05, 1.0)                      # src/config/defaults.py
    assert "large_dataset" in list_presets()
    with pytest.raises(ValueError, match="Unknown preset"):
        get_preset("nope")
    with pytest.raises(FileNotFoundError):
        WakewordConfig.load(tmp_path / "missing.yaml")


def test_model_and_loss_factories_fail_loudly_without_gpu():
    from wakeword_trainer_home_amd.models import create_model, create_loss_function
    from wakeword_trainer_home_amd._native import NativeError
    m = create_model("CNN_Small", num_classes=2, pretrained=False, dropout=0.25)
    from oracle.cnn_small import CNNSmallOracle
    assert list(m.state_dict()) == list(CNNSmallOracle().state_dict())      # interchangeable checkpoints
    assert sum(p.numel() for p in m.parameters()) == 20546
    with pytest.raises(NativeError, match="no CPU fallback"):
        m(torch.zeros(2, 1, 40, 151))
    with pytest.raises(ValueError, match="Unknown architecture"):
        create_model("vit")
    with pytest.raises(ValueError, match="outside this build"):
        create_model("resnet18")
    for arch in ("mobilenetv3", "gru", "crnn"):                      # built, and just as loud off the GPU
        mm = create_model(arch)
        with pytest.raises(NativeError, match="no CPU fallback"):
            mm(torch.zeros(2, 1, 40, 151))
    with pytest.raises(ValueError):
        create_model("cnn_small", num_classes=3)
    crit = create_loss_function("cross_entropy", label_smoothing=0.05, device="cpu")
    assert type(crit).__name__ == "LabelSmoothingCrossEntropy"
    assert type(create_loss_function("cross_entropy", label_smoothing=0.0)).__name__ == "CrossEntropyLoss"
    assert type(create_loss_function("FOCAL_LOSS")).__name__ == "FocalLoss"
    with pytest.raises(ValueError, match="Unknown loss function"):
        create_loss_function("hinge")
    with pytest.raises(ValueError, match="Label smoothing"):
        create_loss_function("cross_entropy", label_smoothing=1.5)
    with pytest.raises(ValueError, match="2D"):
        crit(torch.zeros(4), torch.zeros(4, dtype=torch.long))
    with pytest.raises(ValueError, match="Batch size mismatch"):
        crit(torch.zeros(4, 2), torch.zeros(3, dtype=torch.long))
    with pytest.raises(NativeError, match="no CPU fallback"):
        crit(torch.zeros(4, 2), torch.zeros(4, dtype=torch.long))


def test_library_exports_the_whole_c_abi():
    """libwwhip.so loads and exports every function include/wwhip.h declares (no compute without a GPU)."""
    from wakeword_trainer_home_amd import _native
    header = (REPO / "include" / "wwhip.h").read_text()
    declared = set(re.findall(r"\b(ww_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(str(_native.lib_path()))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    nat_lib = _native.load()
    assert nat_lib.ww_abi_version() == _native.ABI_VERSION
    assert _native.num_frames(24000, 160) == 151
    assert ctypes.sizeof(_native.StepStats) == 48
    # host-side pieces of the ABI follow the oracle's integer law
    from oracle.philox import philox4x32_10, prob_threshold
    assert _native.philox([1, 2, 3, 4], [5, 6]) == [int(v) for v in philox4x32_10(np.array([1, 2, 3, 4], np.uint64),
                                                                                  np.array([5, 6], np.uint64))]
    for p in (0.0, 0.3, 0.5, 1.0):
        assert _native.prob_threshold(float(np.float32(p))) == prob_threshold(p)


@pytest.mark.parametrize("kw", [dict(), dict(n_mels=64), dict(n_mels=128), dict(n_mels=13), dict(n_mels=3), dict(n_mels=80, f_min=20.0),
                                dict(n_mels=23, f_max=3800.0), dict(n_mels=1), dict(n_mels=40, sample_rate=8000),
                                dict(n_mels=40, n_fft=512), dict(n_mels=64, n_fft=256), dict(n_mels=40, n_fft=2048)])
def test_mel_tables_match_the_oracle_filterbank_in_both_forms(kw):
    """Host side of the log-mel kernel (no GPU): the compact HTK bands equal the oracle's filterbank, and the matrix-pipe form of
    the 1024-point kernel (16 blocks of 4 bands per v_mfma_f32_4x4x1, units of bins balanced over the passes) holds every
    non-zero weight exactly once, a quarter of its value, where the kernel's addressing can reach it."""
    from oracle import features as OF
    from wakeword_trainer_home_amd import _native as nat
    cfg = nat.make_feat_cfg(**kw)
    t = nat.mel_tables(cfg)
    M, n_bins = cfg.n_mels, cfg.n_fft // 2 + 1
    fb = OF.mel_filterbank(n_bins, M, cfg.sample_rate, cfg.f_min, cfg.f_max if cfg.f_max > 0 else None)   # (bins, M) float64
    got = np.zeros((n_bins, M), np.float32)
    off = 0
    for m in range(M):
        s_, l_ = int(t["start"][m]), int(t["len"][m])
        got[s_:s_ + l_, m] = t["w"][off:off + l_]
        off += l_
    assert off == t["w"].size
    assert np.array_equal(got, fb.astype(np.float32)), np.abs(got - fb).max()
    tab, qw = t["melq_tab"], t["melq_w"]
    if cfg.n_fft > 1024:                                                                   # the general kernel reads the compact bands
        assert qw.size == 0 and not tab.any()
        return
    # a shorter transform runs on the 1024-point kernel: its bin k is bin r k there, the bins between weigh nothing
    r, n_bins = 1024 // cfg.n_fft, 513
    wide = np.zeros((n_bins, M), np.float32)
    wide[::r] = got
    got = wide
    P, NQ = int(tab[0]), int(tab[1])
    assert NQ == (M + 3) // 4 and (NQ + 15) // 16 <= P <= 4
    first = tab[138:138 + NQ + 1]
    assert first[0] == 0 and first[-1] == 16 * P and np.all(np.diff(first) >= 1)         # every quad owns at least one unit
    quad_of = np.repeat(np.arange(NQ), np.diff(first))
    rec = np.zeros((n_bins + 1024, 4 * NQ), np.float64)
    seen, end = set(), 0
    for p in range(P):
        steps, woff = int(tab[2 + 2 * p]), int(tab[3 + 2 * p])
        assert steps % 8 == 0 and woff == end
        end = woff + 64 * steps
        blk = qw[woff:end].reshape(steps, 16, 4)                                          # (step, block, band in the quad)
        for b in range(16):
            j0, u = int(tab[10 + 32 * p + 2 * b]), int(tab[11 + 32 * p + 2 * b])
            assert j0 % 8 == 0 and 0 <= u < 16 * P and u not in seen
            seen.add(u)
            live = np.flatnonzero(np.abs(blk[:, b, :]).sum(axis=1))
            if live.size:                                                                 # the kernel clamps a group's first bin to 512
                assert j0 + int(live[-1]) <= 512 and (j0 + 8 * (int(live[-1]) // 8)) <= 512
                q = int(quad_of[u])
                rec[j0:j0 + steps, 4 * q:4 * q + 4] += blk[:, b, :]
    assert end == qw.size and len(seen) == 16 * P
    assert np.array_equal((4.0 * rec[:n_bins, :M]).astype(np.float32), got) and not rec[n_bins:].any() and not rec[:, M:].any()
    # units of a quad that hold bins: at most 8 (the kernel fetches a band's partials as eight loads), in bin order, ahead of the
    # empty ones
    for q in range(NQ):
        us = list(range(int(first[q]), int(first[q + 1])))
        bins = []
        for u in us:
            p, b = next((p, b) for p in range(P) for b in range(16) if int(tab[11 + 32 * p + 2 * b]) == u)
            steps, woff = int(tab[2 + 2 * p]), int(tab[3 + 2 * p])
            w_u = qw[woff:woff + 64 * steps].reshape(steps, 16, 4)[:, b, :]
            live = np.flatnonzero(np.abs(w_u).sum(axis=1))
            bins.append((int(tab[10 + 32 * p + 2 * b]) + int(live[0])) if live.size else None)
        filled = [x for x in bins if x is not None]
        assert len(filled) <= 8 and bins[:len(filled)] == filled and filled == sorted(filled), (q, bins)


def test_flat_buckets_mixin_on_cpu():
    """models/flat_buckets.py (device-agnostic host logic): parameters become views of one flat tensor in parameters()
    order, state_dict round-trips through the views, gather_grads copies every gradient and refuses a missing one (torch.optim
    would skip that parameter; the bucketed step cannot), _apply rebuilds."""
    import torch.nn as nn
    from wakeword_trainer_home_amd.models.flat_buckets import FlatBuckets

    class Net(FlatBuckets, nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(3, 4), nn.Linear(4, 2)

    torch.manual_seed(0)
    net = Net()
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    flat = net.flat_param
    assert flat.numel() == sum(p.numel() for p in net.parameters()) == 3 * 4 + 4 + 4 * 2 + 2
    off = 0
    for p in net.parameters():                                     # views, in order, values preserved
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    assert all(torch.equal(v, ref[k]) for k, v in net.state_dict().items())
    flat.mul_(2.0)                                                 # an update of the bucket is an update of the model
    assert torch.equal(net.a.weight, 2 * ref["a.weight"])
    net.load_state_dict(ref)                                       # ... and load_state_dict writes through the views
    assert torch.equal(flat[:12].view(4, 3), ref["a.weight"]) and net.flat_param.data_ptr() == flat.data_ptr()
    out = net.b(torch.relu(net.a(torch.ones(5, 3)))).sum()
    out.backward()
    net.flat_grad.fill_(7.0)
    g = net.gather_grads()
    assert torch.equal(g[:12].view(4, 3), net.a.weight.grad) and torch.equal(g[-2:], net.b.bias.grad)
    assert not net.grads_in_bucket()
    net.b.bias.grad = None                                         # a parameter without gradient: refused, not zero-filled
    with pytest.raises(RuntimeError, match="no gradient"):
        net.gather_grads()
    net.to("cpu", torch.float32)                                   # _apply -> rebuilt lazily
    assert net._fb_param is None and net.flat_param.numel() == flat.numel()
    import copy
    twin = copy.deepcopy(net)                                      # tensors are copied one by one: the twin rebuilds its bucket
    assert twin.flat_param.data_ptr() != net.flat_param.data_ptr()
    assert twin.a.weight.data_ptr() == twin.flat_param.data_ptr() and torch.equal(twin.flat_param, net.flat_param)
    with pytest.raises(ValueError, match="float32"):
        Net().double().flat_param


def test_bench_self_launch_spawns_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (torch.distributed.run children).  Here, without
    a GPU, each child must refuse loudly (no CPU fallback) and the parent must relay the failure: non-zero exit, nothing on
    stdout -- never a JSON line from a path that did not run the HIP kernels."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs an MI355X" in r.stderr


def test_flat_bucket_gradient_slots_are_adopted_by_autograd_and_owner_checked():
    """models/flat_buckets.py: a backward function that writes a parameter gradient into ``grad_slot(param)`` and returns that view
    has it adopted as ``param.grad`` without a copy (``gather_grads`` then moves nothing); a deep copy of the model does not
    inherit the original's slots; a second backward without zero_grad accumulates instead of overwriting."""
    import copy
    import torch.nn as nn
    from wakeword_trainer_home_amd.models.flat_buckets import FlatBuckets, grad_slot

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            ctx.w = w
            return x @ w.T

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            dw, out = g.T @ x, grad_slot(ctx.w)
            if out is not None:
                out.copy_(dw)
                dw = out
            return g @ ctx.w, dw

    class M(FlatBuckets, nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Parameter(torch.randn(3, 4)), nn.Parameter(torch.randn(2, 3))

        def forward(self, x):
            return Fn.apply(Fn.apply(x, self.a), self.b)

    torch.manual_seed(0)
    m, x = M(), torch.randn(5, 4)
    _ = m.flat_grad                                              # builds the bucket and the slots
    m(x).sum().backward()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(m._fb_plist, m._fb_views))
    ref = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    assert torch.equal(m.gather_grads(), ref)
    m(x).sum().backward()                                        # no zero_grad: the slot is taken, autograd accumulates
    assert torch.allclose(m.flat_grad, 2 * ref)
    twin = copy.deepcopy(m)
    assert grad_slot(twin.a) is None                             # no bucket of its own yet, and never the original's
    _ = twin.flat_grad
    s = grad_slot(twin.a)
    assert s.data_ptr() == twin._fb_views[0].data_ptr() != m._fb_views[0].data_ptr()


def test_pending_batch_counts_do_not_survive_a_load_and_modules_pickle():
    """BatchNorm's num_batches_tracked is counted on the host between state reads (no kernel per layer per step).  Loading a
    state dict makes the loaded value the truth -- a count pending from earlier steps must not be added to it -- and the hooks
    are plain functions, so whole-module pickling works."""
    import pickle
    from wakeword_trainer_home_amd.models.mobilenet import _BN
    bn = _BN(4)
    bn._pending_tracked = 3
    assert int(bn.state_dict()["num_batches_tracked"]) == 3 and bn._pending_tracked == 0
    bn._pending_tracked = 5                                   # steps after the save ...
    sd = {k: v.clone() for k, v in bn.state_dict().items()}   # (flushes: 8)
    sd["num_batches_tracked"] = torch.tensor(10)
    bn._pending_tracked = 2                                   # ... and more steps, then a load in the same process
    bn.load_state_dict(sd)
    assert bn._pending_tracked == 0 and int(bn.state_dict()["num_batches_tracked"]) == 10
    clone = pickle.loads(pickle.dumps(bn))
    clone._pending_tracked = 1
    assert int(clone.state_dict()["num_batches_tracked"]) == 11


@pytest.mark.timeout(300)
def test_bench_spawner_starts_its_ranks_and_fails_loudly_without_a_gpu():
    """``python bench.py --gpus 2`` with no launcher: the parent starts its own two ranks (torch.distributed.run, before anything
    touches HIP) and relays rank 0's line.  Off an MI355X every rank must refuse loudly -- the hot path has no CPU fallback -- so
    the parent exits non-zero with NOTHING on stdout (a consumer parsing one JSON line sees none) and the reason on stderr.
    (The same path with two real ranks on the GPU: tests/test_data_parallel_gpu.py.)"""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the N-rank command path is covered by the -m gpu suite")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env, timeout=280)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs an MI355X" in r.stderr
