"""Generate the golden fixtures by IMPORTING THE REFERENCE (build container only).

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Writes small data-only fixtures next to this file.  The GPU box never sees
/root/reference; tests read only the committed .npz/.json files.

What is captured:
  G1  src.models.losses.create_loss_function  -> loss + autograd dlogits
  G2  src.training.trainer.Trainer (enforce_cuda patched, device='cpu') driving the
      build's cnn_small oracle: per-step loss/acc/grad-norm, LR per epoch, history,
      final parameters
  G3  src.training.metrics.MetricsCalculator -> MetricResults.to_dict()
  G4  src.training.optimizer_factory schedulers under Trainer._update_scheduler's
      calling convention (scheduler.step(val_loss))  -> LR trajectories
  G5  Trainer._save_checkpoint -> key set / filenames / TrainingState fields
"""
import json
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REF))
sys.path.insert(0, str(REPO))
warnings.filterwarnings("ignore")

from src.models.losses import create_loss_function            # noqa: E402  (reference)
import src.training.trainer as ref_trainer                     # noqa: E402  (reference)
from src.training.metrics import MetricsCalculator             # noqa: E402  (reference)
from src.training import optimizer_factory as ref_of           # noqa: E402  (reference)
from src.config.defaults import WakewordConfig                 # noqa: E402  (reference)

from oracle.cnn_small import CNNSmallOracle                    # noqa: E402


# ----------------------------------------------------------------------------- G1
def g1_loss():
    out = {}
    g = torch.Generator().manual_seed(101)
    cases = {
        "b512": (torch.randn(512, 2, generator=g) * 2.0, torch.randint(0, 2, (512,), generator=g)),
        "b7": (torch.randn(7, 2, generator=g), torch.randint(0, 2, (7,), generator=g)),
        "extreme": (torch.tensor([[30.0, -30.0], [-30.0, 30.0], [30.0, -30.0], [-30.0, 30.0],
                                  [0.0, 0.0], [88.0, -88.0], [1e-3, -1e-3], [-12.5, 12.25]]),
                    torch.tensor([0, 0, 1, 1, 1, 1, 0, 0])),
    }
    specs = [("cross_entropy", dict(label_smoothing=e)) for e in (0.0, 0.05, 0.1, 0.15)]
    specs += [("focal_loss", dict(focal_alpha=0.25, focal_gamma=2.0)),
              ("focal_loss", dict(focal_alpha=0.25, focal_gamma=2.5)),
              ("focal_loss", dict(focal_alpha=0.75, focal_gamma=0.0))]
    for cname, (z, y) in cases.items():
        out[f"{cname}/logits"] = z.numpy()
        out[f"{cname}/targets"] = y.numpy()
        for i, (name, kw) in enumerate(specs):
            crit = create_loss_function(name, num_classes=2, device="cpu", **kw)
            zz = z.clone().requires_grad_(True)
            loss = crit(zz, y)
            (gz,) = torch.autograd.grad(loss, zz)
            out[f"{cname}/spec{i}/loss"] = np.float64(loss.item())
            out[f"{cname}/spec{i}/dlogits"] = gz.numpy()
    out["specs"] = np.array(json.dumps(specs))
    np.savez_compressed(HERE / "g1_loss.npz", **out)
    print("G1 written")


# ----------------------------------------------------------------------------- G2 / G5
def make_inputs(seed, n, F=40, T=151):
    """log-mel-like synthetic features; regenerated from the seed in the tests."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, F, T, generator=g) * 2.0 - 4.0
    y = (torch.rand(n, generator=g) < 0.3).long()
    return x, y


class Recorder:
    def __init__(self):
        self.loss, self.acc, self.epochs = [], [], []

    def on_batch_end(self, batch_idx, loss, acc):
        self.loss.append(loss)
        self.acc.append(acc)

    def on_epoch_end(self, epoch, train_loss, val_loss, val_metrics):
        self.epochs.append(dict(epoch=epoch, train_loss=train_loss, val_loss=val_loss,
                                val_metrics=val_metrics.to_dict()))


def run_trace(tag, cfg_mut, batch, n_train, n_val, epochs, out, meta):
    ref_trainer.enforce_cuda = lambda: None          # the CUDA gate
    torch.manual_seed(1234)
    model = CNNSmallOracle(dropout=0.0)
    init_sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = WakewordConfig()
    cfg.model.architecture = "cnn_small"
    cfg.training.epochs = epochs
    cfg.training.batch_size = batch
    cfg.training.checkpoint_frequency = "every_epoch"
    cfg_mut(cfg)
    xtr, ytr = make_inputs(7001, n_train)
    xva, yva = make_inputs(7002, n_val)
    dl = torch.utils.data.DataLoader
    tr = dl(torch.utils.data.TensorDataset(xtr, ytr), batch_size=batch, shuffle=False)
    va = dl(torch.utils.data.TensorDataset(xva, yva), batch_size=batch, shuffle=False)
    gnorms = []
    orig_clip = ref_of.clip_gradients

    def rec_clip(m, max_norm, norm_type=2.0):
        v = orig_clip(m, max_norm, norm_type)
        gnorms.append(v)
        return v
    ref_trainer.clip_gradients = rec_clip
    with tempfile.TemporaryDirectory() as d:
        t = ref_trainer.Trainer(model, tr, va, cfg, checkpoint_dir=Path(d), device="cpu")
        rec = Recorder()
        t.add_callback(rec)
        res = t.train()
        files = sorted(p.name for p in Path(d).iterdir())
        ck = torch.load(Path(d) / files[0], map_location="cpu", weights_only=False)
        meta[tag] = dict(
            files=files, ckpt_keys=sorted(ck.keys()),
            state_fields=sorted(vars(ck["state"]).keys()),
            criterion=type(t.criterion).__name__, optimizer=type(t.optimizer).__name__,
            scheduler=type(t.scheduler).__name__ if t.scheduler else None,
            batch=batch, n_train=n_train, n_val=n_val, epochs=epochs,
            train_seed=7001, val_seed=7002,
            history={k: [float(v) for v in vs] for k, vs in res["history"].items()},
            final_epoch=res["final_epoch"], best_val_loss=res["best_val_loss"],
            best_val_f1=res["best_val_f1"], best_val_fpr=res["best_val_fpr"],
            best_f1_epoch=res["best_f1_epoch"], best_fpr_epoch=res["best_fpr_epoch"],
            epochs_rec=rec.epochs, global_step=t.state.global_step,
            cfg=dict(loss=cfg.loss.to_dict(), optimizer=cfg.optimizer.to_dict(),
                     training=cfg.training.to_dict()))
    ref_trainer.clip_gradients = orig_clip
    out[f"{tag}/step_loss"] = np.array(rec.loss, dtype=np.float64)
    out[f"{tag}/step_acc"] = np.array(rec.acc, dtype=np.float64)
    out[f"{tag}/grad_norm"] = np.array(gnorms, dtype=np.float64)
    for k, v in init_sd.items():
        out[f"{tag}/init/{k}"] = v.numpy()
    for k, v in model.state_dict().items():
        out[f"{tag}/final/{k}"] = v.numpy()
    print(tag, "losses", rec.loss[:4], "...", "lr", meta[tag]["history"]["learning_rates"])


def g2_trace():
    out, meta = {}, {}

    def default(cfg):
        pass

    def small(cfg):                       # small-dataset preset's loss/optim knobs (presets.py:35-92)
        cfg.loss.loss_function = "focal_loss"
        cfg.loss.focal_alpha = 0.25
        cfg.loss.focal_gamma = 2.0
        cfg.training.learning_rate = 0.0005
        cfg.optimizer.scheduler = "plateau"
        cfg.optimizer.warmup_epochs = 0
        cfg.optimizer.gradient_clip = 0.5

    def sgd(cfg):
        cfg.optimizer.optimizer = "sgd"
        cfg.optimizer.scheduler = "step"
        cfg.optimizer.step_size = 1
        cfg.optimizer.gamma = 0.5
        cfg.optimizer.warmup_epochs = 0
        cfg.loss.label_smoothing = 0.0
        cfg.training.learning_rate = 0.01

    run_trace("default_b16", default, 16, 96, 32, 4, out, meta)
    run_trace("focal_b16", small, 16, 64, 32, 2, out, meta)
    run_trace("sgd_b8", sgd, 8, 40, 16, 3, out, meta)
    run_trace("default_b128", default, 128, 256, 128, 4, out, meta)
    np.savez_compressed(HERE / "g2_trace.npz", **out)
    (HERE / "g2_meta.json").write_text(json.dumps(meta, indent=1))
    print("G2/G5 written")


# ----------------------------------------------------------------------------- G3
def g3_metrics():
    g = torch.Generator().manual_seed(303)
    cases = {}
    for name, n, pos in (("mixed", 300, 0.3), ("allneg", 50, 0.0), ("allpos", 20, 1.0), ("one", 1, 0.5)):
        z = torch.randn(n, 2, generator=g)
        y = (torch.rand(n, generator=g) < pos).long()
        if name == "mixed":
            z[:10] = 0.0                   # argmax ties -> class 0
        m = MetricsCalculator(device="cpu").calculate(z, y)
        cases[name] = dict(logits=z.tolist(), targets=y.tolist(), result=m.to_dict())
    (HERE / "g3_metrics.json").write_text(json.dumps(cases))
    print("G3 written")


# ----------------------------------------------------------------------------- G4
def g4_sched():
    out = {}
    val_losses = [0.9, 0.69, 0.66, 0.70, 0.71, 0.72, 0.5, 3.5, 0.4, 0.41, 0.42, 0.43]
    specs = {
        "cosine_warm3": dict(scheduler="cosine", warmup_epochs=3),
        "cosine_nowarm": dict(scheduler="cosine", warmup_epochs=0),
        "step": dict(scheduler="step", warmup_epochs=0, step_size=3, gamma=0.5),
        "step_warm2": dict(scheduler="step", warmup_epochs=2, step_size=3, gamma=0.5),
        "plateau": dict(scheduler="plateau", warmup_epochs=0, patience=2, factor=0.5, min_lr=1e-5),
        "plateau_warm1": dict(scheduler="plateau", warmup_epochs=1, patience=2, factor=0.5, min_lr=1e-5),
        "none": dict(scheduler="none", warmup_epochs=0),
    }
    for name, kw in specs.items():
        cfg = WakewordConfig()
        cfg.training.epochs = len(val_losses)
        for k, v in kw.items():
            setattr(cfg.optimizer, k, v)
        model = torch.nn.Linear(4, 2)
        opt, sch = ref_of.create_optimizer_and_scheduler(model, cfg)
        # duck-typed holder so the reference's own _update_scheduler body runs unchanged
        holder = type("H", (), {})()
        holder.scheduler = sch
        lrs = [ref_of.get_learning_rate(opt)]
        for vl in val_losses:
            opt.step()
            ref_trainer.Trainer._update_scheduler(holder, vl)
            lrs.append(ref_of.get_learning_rate(opt))
        out[name] = dict(cfg=kw, lrs=lrs)
    out["val_losses"] = val_losses
    (HERE / "g4_sched.json").write_text(json.dumps(out, indent=1))
    print("G4 written")


# ----------------------------------------------------------------------------- G7
def g7_gru():
    """The reference's own GRUWakeword (src/models/architectures.py:198-267).  Its module imports torchvision at the top;
    torchvision is not installed here, so an empty stand-in module satisfies that import -- GRUWakeword itself only uses
    torch.nn.  eval-mode logits (dropout off -> deterministic) and train-mode gradients with dropout=0."""
    import types
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.models = types.ModuleType("torchvision.models")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tv.models
    from src.models.architectures import create_model as ref_create_model       # noqa: E402  (reference)
    torch.manual_seed(77)
    model = ref_create_model("gru", num_classes=2, input_size=40, hidden_size=128, num_layers=2, bidirectional=True,
                             dropout=0.0)
    g = torch.Generator().manual_seed(78)
    x = torch.randn(6, 31, 40, generator=g)
    y = torch.tensor([0, 1, 1, 0, 1, 0])
    model.eval()
    with torch.no_grad():
        logits_eval = model(x)
    model.train()
    out = model(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    grads = {"grad." + k: p.grad.numpy() for k, p in model.named_parameters()}
    np.savez_compressed(HERE / "g7_gru.npz", x=x.numpy(), y=y.numpy(), logits_eval=logits_eval.numpy(),
                        logits_train=out.detach().numpy(), loss=np.float64(loss.item()), **{"sd." + k: v for k, v in sd.items()},
                        **grads)
    print("g7_gru.npz:", list(sd.keys()), "loss", loss.item())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        g7_gru()
        sys.exit(0)
    g1_loss()
    g3_metrics()
    g4_sched()
    g2_trace()
