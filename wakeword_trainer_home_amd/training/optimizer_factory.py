"""Optimizer / scheduler / scaler glue -- stays ordinary PyTorch host code (north_star: "Python host
code ... for the outer DataLoader/optimizer glue").  Behaviour follows the reference factory
(``src/training/optimizer_factory.py``): optimizer choice and arguments (:121-209, SGD is Nesterov),
scheduler construction with ``T_max = epochs - warmup`` (:281-333), the ``WarmupScheduler`` wrapper
whose ``step(epoch=None, metrics=None)`` signature produces quirk Q2 when the Trainer calls
``scheduler.step(val_loss)`` (:56-85; SURVEY.md Q2, pinned by tests/golden/g4_sched.json), the
argument validation messages, ``GradScaler`` (:403-420) and ``clip_gradients`` (:423-452).
For the HIP-backed model the clip runs on the flat gradient bucket in one kernel (ww_grad_norm_clip).
"""
import logging
from typing import Any, Optional, Tuple

import torch
import torch.nn as nn
import torch.optim as optim
from torch.optim.lr_scheduler import CosineAnnealingLR, ReduceLROnPlateau, StepLR

logger = logging.getLogger(__name__)


class WarmupScheduler:
    """Linear LR warm-up over ``warmup_epochs``, then delegate to ``base_scheduler``."""

    def __init__(self, optimizer: optim.Optimizer, warmup_epochs: int, base_scheduler: Optional[Any] = None):
        if warmup_epochs < 0:
            raise ValueError(f"warmup_epochs must be non-negative, got {warmup_epochs}")
        self.optimizer, self.warmup_epochs, self.base_scheduler = optimizer, warmup_epochs, base_scheduler
        self.current_epoch = 0
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        if any(lr <= 0 for lr in self.base_lrs):
            raise ValueError(f"All learning rates must be positive, got {self.base_lrs}")

    def step(self, epoch: Optional[int] = None, metrics: Optional[float] = None):
        # NB: the Trainer passes val_loss POSITIONALLY, so it lands in `epoch` (reference quirk Q2)
        self.current_epoch = epoch if epoch is not None else self.current_epoch + 1
        if self.current_epoch < self.warmup_epochs:
            k = (self.current_epoch + 1) / self.warmup_epochs
            for group, base in zip(self.optimizer.param_groups, self.base_lrs):
                group["lr"] = base * k
        elif self.base_scheduler is not None:
            if isinstance(self.base_scheduler, ReduceLROnPlateau):
                if metrics is not None:
                    self.base_scheduler.step(metrics)
            else:
                self.base_scheduler.step()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"current_epoch": self.current_epoch, "base_lrs": self.base_lrs, "warmup_epochs": self.warmup_epochs,
                "base_scheduler_state": self.base_scheduler.state_dict() if self.base_scheduler else None}

    def load_state_dict(self, state):
        self.current_epoch, self.base_lrs = state["current_epoch"], state["base_lrs"]
        self.warmup_epochs = state["warmup_epochs"]
        if self.base_scheduler and state["base_scheduler_state"]:
            self.base_scheduler.load_state_dict(state["base_scheduler_state"])


class FlatFusedOptimizer(optim.Optimizer):
    """Adam / AdamW / SGD-Nesterov for a HIP-backed model whose parameters and gradients live in flat fp32 buckets
    (``model.flat_param`` / ``model.flat_grad``): gradient clipping and the update are ONE kernel launch
    (``ww_clip_optim_step``), the "skip a non-finite batch" decision (trainer.py:177-179) is taken on the device, and the
    step count lives there too.  Update rules, hyper-parameter names, ``param_groups`` and the ``state_dict()`` layout are
    torch.optim's (``optimizer_factory.py:165-199``), so LR schedulers and reference checkpoints work unchanged."""

    def __init__(self, model: nn.Module, kind: str, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, momentum: float = 0.0):
        from .. import _native as nat
        self._nat = nat
        self._kind = {"adam": nat.OPT_ADAM, "adamw": nat.OPT_ADAMW, "sgd": nat.OPT_SGD}[kind]
        if self._kind == nat.OPT_SGD:
            defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=True, maximize=False,
                            foreach=None, differentiable=False, fused=None)
        else:
            defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                            foreach=None, capturable=False, differentiable=False, fused=None)
        frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
        if frozen:
            # the fused kernel walks the WHOLE bucket (decay and moment updates included); torch.optim skips parameters
            # without a gradient -- create_optimizer() hands such models to torch.optim instead
            raise ValueError(f"FlatFusedOptimizer updates every parameter of the bucket; frozen parameters {frozen[:3]}... "
                             "are not supported (create_optimizer falls back to torch.optim for them)")
        super().__init__(list(model.parameters()), defaults)
        self._model = model
        flat = model.flat_param
        if not flat.is_cuda:
            raise nat.NativeError("FlatFusedOptimizer needs the model on an MI355X ('cuda') device")
        self._flat_ptr = flat.data_ptr()
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat) if self._kind != nat.OPT_SGD else None
        self._step_state = torch.zeros(2, dtype=torch.int64, device=flat.device)
        self._parity = 0
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.device)   # pre-clip norm of the last step

    # ------------------------------------------------------------------ the step
    def _cfg(self, max_norm):
        g = self.param_groups[0]
        b1, b2 = g.get("betas", (0.0, 0.0))
        return self._nat.OptimCfg(self._kind, g["lr"], b1, b2, g.get("eps", 0.0), g["weight_decay"], g.get("momentum", 0.0),
                                  max_norm)

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0, stats=None, stats_host=None, gathered: bool = False,
             found_inf_extra=None, stats_host_alt=None, loss_scale=None):
        """``max_norm > 0`` also clips (clip_grad_norm_ semantics, in place on the bucket); ``stats`` is the step's
        device ``ww_step_stats`` (found_inf gate, grad_norm output); ``stats_host`` a pinned 48-byte tensor the kernel
        copies it to; ``gathered``: ``flat_grad`` already holds this step's (all-reduced) gradients; ``found_inf_extra``:
        float32[1] device tensor, non-zero = skip (the all-reduced verdict of the other ranks); ``stats_host_alt``: second
        pinned buffer, used instead of ``stats_host`` when a bound step control block says parity 1 (graph replay);
        ``loss_scale``: device ``ww_loss_scale`` of the fp16 storage mode (gradients are unscaled first, the scale is
        updated by GradScaler's rule; its slot is this optimizer's parity -- ``scale_slot``)."""
        if closure is not None:
            raise ValueError("FlatFusedOptimizer does not support closures")
        model = self._model
        if model.flat_param.data_ptr() != self._flat_ptr:
            raise self._nat.NativeError("the model's parameter bucket moved after the optimizer was created "
                                        "(model.to(...) / dtype change): create the optimizer afterwards")
        if model.flat_grad is None or all(p.grad is None for p in self.param_groups[0]["params"]):
            return None                                      # nothing to do, like torch's optimizers
        if gathered:                                         # the caller gathered (and all-reduced) flat_grad already
            pass
        elif hasattr(model, "gather_grads"):
            if not model.grads_in_bucket():
                model.gather_grads()
        elif not model.grads_in_bucket():                    # gradients were accumulated outside the bucket: gather
            off = 0
            for p in self.param_groups[0]["params"]:
                n = p.numel()
                if p.grad is None:
                    raise RuntimeError("FlatFusedOptimizer.step(): a parameter has no gradient; torch.optim would skip it, the "
                                       "fused kernel cannot (it updates the whole bucket)")
                model.flat_grad[off:off + n].copy_(p.grad.reshape(-1))
                off += n
        self._nat.clip_optim_step_(self._cfg(max_norm), model.flat_param, model.flat_grad, self._m, self._v,
                                   self._step_state, self._parity, norm_out=self.grad_norm, stats=stats,
                                   stats_host=stats_host, found_inf_extra=found_inf_extra, stats_host_alt=stats_host_alt,
                                   loss_scale=loss_scale)
        self._parity ^= 1
        return None

    @property
    def scale_slot(self) -> int:
        """Slot of ``ww_loss_scale`` the NEXT step reads (the loss kernel of that step must use the same one)."""
        return self._parity

    def step_count(self) -> int:
        """Number of applied (not skipped) updates; synchronises."""
        return int(self._step_state[self._parity].item())

    # ------------------------------------------------------------------ torch-compatible state
    def _views(self, flat):
        out, off = [], 0
        for p in self.param_groups[0]["params"]:
            out.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    def state_dict(self):
        t = self.step_count()
        self.state.clear()
        if t > 0:
            params = self.param_groups[0]["params"]
            if self._kind == self._nat.OPT_SGD:
                mom = self.param_groups[0].get("momentum", 0.0) != 0.0
                for p, m in zip(params, self._views(self._m)):
                    self.state[p] = {"momentum_buffer": m.clone() if mom else None}      # torch.optim.SGD: None without momentum
            else:
                for p, m, v in zip(params, self._views(self._m), self._views(self._v)):
                    self.state[p] = {"step": torch.tensor(float(t)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        sd = super().state_dict()
        self.state.clear()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        params = self.param_groups[0]["params"]
        t = 0
        self._m.zero_()
        if self._v is not None:
            self._v.zero_()
        for p, m, v in zip(params, self._views(self._m), self._views(self._v) if self._v is not None else [None] * len(params)):
            st = self.state.get(p)
            if not st:
                continue
            if "momentum_buffer" in st and st["momentum_buffer"] is not None:
                m.copy_(st["momentum_buffer"])
                t = max(t, 1)
            if "exp_avg" in st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                t = max(t, int(float(st["step"])))
        self.state.clear()
        self._step_state.fill_(t)


def create_optimizer(model: nn.Module, optimizer_name: str = "adam", learning_rate: float = 0.001,
                     weight_decay: float = 1e-4, momentum: float = 0.9, betas: Tuple[float, float] = (0.9, 0.999),
                     **kwargs) -> optim.Optimizer:
    if learning_rate <= 0:
        raise ValueError(f"Learning rate must be positive, got {learning_rate}")
    if weight_decay < 0:
        raise ValueError(f"Weight decay must be non-negative, got {weight_decay}")
    if not 0 <= momentum <= 1:
        raise ValueError(f"Momentum must be in [0, 1], got {momentum}")
    if not all(0 <= b <= 1 for b in betas):
        raise ValueError(f"Betas must be in [0, 1], got {betas}")
    name = optimizer_name.lower()
    params = list(model.parameters())
    if (name in ("adam", "adamw", "sgd") and not kwargs and hasattr(model, "flat_param") and params
            and all(p.is_cuda and p.requires_grad for p in params)):
        # HIP-backed model: clip + update in one launch on the flat buckets (same update rules, same state_dict layout)
        return FlatFusedOptimizer(model, name, learning_rate, betas=betas, weight_decay=weight_decay, momentum=momentum)
    if name in ("adam", "adamw") and "fused" not in kwargs and params and all(p.is_cuda for p in params):
        kwargs["fused"] = True        # one multi-tensor kernel per step instead of ~9 (same update rule)
    if name == "adam":
        return optim.Adam(params, lr=learning_rate, betas=betas, weight_decay=weight_decay, **kwargs)
    if name == "adamw":
        return optim.AdamW(params, lr=learning_rate, betas=betas, weight_decay=weight_decay, **kwargs)
    if name == "sgd":
        return optim.SGD(params, lr=learning_rate, momentum=momentum, weight_decay=weight_decay, nesterov=True, **kwargs)
    raise ValueError(f"Unknown optimizer: {name}. Supported: adam, adamw, sgd")


def create_scheduler(optimizer: optim.Optimizer, scheduler_name: str = "cosine", epochs: int = 50,
                     warmup_epochs: int = 0, step_size: int = 10, gamma: float = 0.1, patience: int = 5,
                     factor: float = 0.5, min_lr: float = 1e-6, **kwargs) -> Optional[Any]:
    checks = ((epochs > 0, f"Epochs must be positive, got {epochs}"),
              (warmup_epochs >= 0, f"Warmup epochs must be non-negative, got {warmup_epochs}"),
              (warmup_epochs < epochs, f"Warmup epochs ({warmup_epochs}) must be less than total epochs ({epochs})"),
              (step_size > 0, f"Step size must be positive, got {step_size}"),
              (0 < gamma <= 1, f"Gamma must be in (0, 1], got {gamma}"),
              (patience > 0, f"Patience must be positive, got {patience}"),
              (0 < factor < 1, f"Factor must be in (0, 1), got {factor}"),
              (min_lr >= 0, f"Minimum LR must be non-negative, got {min_lr}"))
    for ok, msg in checks:
        if not ok:
            raise ValueError(msg)
    name = scheduler_name.lower()
    if name == "none":
        return None
    if name == "cosine":
        base = CosineAnnealingLR(optimizer, T_max=epochs - warmup_epochs if warmup_epochs > 0 else epochs,
                                 eta_min=min_lr, **kwargs)
    elif name == "step":
        base = StepLR(optimizer, step_size=step_size, gamma=gamma, **kwargs)
    elif name == "plateau":
        base = ReduceLROnPlateau(optimizer, mode="min", factor=factor, patience=patience, min_lr=min_lr, **kwargs)
    else:
        raise ValueError(f"Unknown scheduler: {name}. Supported: cosine, step, plateau, none")
    return WarmupScheduler(optimizer, warmup_epochs, base) if warmup_epochs > 0 else base


def create_optimizer_and_scheduler(model: nn.Module, config: Any) -> Tuple[optim.Optimizer, Optional[Any]]:
    o, t = config.optimizer, config.training
    optimizer = create_optimizer(model, o.optimizer, t.learning_rate, o.weight_decay, o.momentum, tuple(o.betas))
    scheduler = create_scheduler(optimizer, o.scheduler, t.epochs, o.warmup_epochs, o.step_size, o.gamma, o.patience,
                                 o.factor, o.min_lr)
    return optimizer, scheduler


def get_learning_rate(optimizer: optim.Optimizer) -> float:
    return optimizer.param_groups[0]["lr"]


def adjust_learning_rate(optimizer: optim.Optimizer, scale: float):
    for group in optimizer.param_groups:
        group["lr"] *= scale


def create_grad_scaler(enabled: bool = True):
    return torch.amp.GradScaler("cuda", enabled=enabled)


def clip_gradients(model: nn.Module, max_norm: float, norm_type: float = 2.0) -> float:
    """Generic-module path (syncs, like the reference).  HIP-backed models are clipped on their flat
    bucket by the Trainer without a separate sync."""
    if max_norm <= 0:
        raise ValueError(f"max_norm must be positive, got {max_norm}")
    if norm_type <= 0:
        raise ValueError(f"norm_type must be positive, got {norm_type}")
    return torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm, norm_type=norm_type).item()


class DeviceGradScaler:
    """What ``trainer.scaler`` is in the fp16 storage mode: torch.amp.GradScaler's state (scale, growth tracker and its
    three hyper-parameters: create_grad_scaler, src/training/optimizer_factory.py:403-420) living in a device
    ``ww_loss_scale`` that the loss kernel reads and the fused optimizer updates -- scale / unscale_ / step / update
    (src/training/trainer.py:182-193) without a host round trip.  ``state_dict()`` has GradScaler's keys, so the
    checkpoint's ``scaler_state_dict`` entry keeps its schema."""

    def __init__(self, device, optimizer, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        from .. import _native as nat
        self._nat, self._opt = nat, optimizer
        self.state = nat.loss_scale_new(device, init_scale, growth_factor, backoff_factor, growth_interval)

    def is_enabled(self) -> bool:
        return True

    def _read(self):
        return self._nat.loss_scale_read(self.state, self._opt.scale_slot)

    def get_scale(self) -> float:
        return float(self._read()["scale"])

    def state_dict(self):
        r = self._read()
        return {"scale": float(r["scale"]), "growth_factor": r["growth_factor"], "backoff_factor": r["backoff_factor"],
                "growth_interval": r["growth_interval"], "_growth_tracker": int(r["growth_tracker"])}

    def load_state_dict(self, sd):
        if not sd:
            return
        self.state.copy_(self._nat.loss_scale_new("cpu", sd["scale"], sd["growth_factor"], sd["backoff_factor"],
                                                  sd["growth_interval"], sd.get("_growth_tracker", 0)))
