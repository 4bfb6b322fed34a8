"""Training loop with the reference ``Trainer`` API (``src/training/trainer.py:44-585``) and an
MI355X-native inner step.

Kept from the reference (pinned by tests/golden/g2_*.{npz,json} captured from the real class):
constructor signature and attributes (:50-126), ``train()`` result dict and history keys (:328-337,
:409-418), per-epoch order train -> validate -> scheduler(val_loss) -> history -> improvement ->
checkpoint -> early stop -> ``on_epoch_end`` (:350-379), the "any of loss/F1/FPR improved" rule
(:431-457), checkpoint schema, file names and atomic tmp+rename (:463-525), resume (:527-575),
callback protocol ``on_epoch_start / on_batch_end(batch_idx, loss, acc) / on_epoch_end`` (:577-585),
batch contract (2- or 3-tuples, empty tensors skipped, :150-162), error policy (OOM -> skip, other
RuntimeError -> re-raise, other Exception -> log + skip, :214-226), non-finite loss -> batch skipped
but still counted in the epoch-loss denominator (:177-179, :229).

Replaced (the hot path, :165-203): when the model is HIP-backed the step is
``frontend (optional) -> ww_cnn_small_fwd -> ww_ce2_loss_fwd_bwd -> ww_cnn_small_bwd ->
[all-reduce of the flat gradient bucket] -> ww_grad_norm_clip -> optimizer.step`` with ONE 48-byte
device->host read per step (loss, accuracy counters, finite flag, grad norm) instead of the
reference's >= 6 synchronisations, and none of them before ``optimizer.step()``: the "skip this batch" decision
(non-finite loss, invalid targets, non-finite gradient norm) reaches the fused optimizer as a device flag.  With
``config.training.deferred_metrics`` (default on for the HIP model) a step's stats are read while the next step
runs, so ``on_batch_end`` fires one step late (same order, same values; flushed at epoch end); set it to False for
the reference's strict timing.  Extension: 2-D inputs ``(B, N)`` are raw waveforms and go
through the fused log-mel + SpecAugment kernel first.  Any other ``nn.Module`` takes the
reference's autograd step unchanged, so the class stays a drop-in.
"""
import logging
import sys
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from .. import _native as nat
from ..config.cuda_utils import enforce_cuda
from ..models.architectures import CNNSmallWakeword
from ..models.losses import create_loss_function, _NativeLoss
from .metrics import MetricMonitor, MetricResults, MetricsTracker
from .optimizer_factory import (FlatFusedOptimizer, clip_gradients, create_grad_scaler,
                                create_optimizer_and_scheduler, get_learning_rate)

logger = logging.getLogger(__name__)

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


@dataclass
class TrainingState:
    epoch: int = 0
    global_step: int = 0
    best_val_loss: float = float("inf")
    best_val_f1: float = 0.0
    best_val_fpr: float = 1.0
    epochs_without_improvement: int = 0
    training_time: float = 0.0


def _checkpoint_safe_globals():
    """Classes a checkpoint may contain besides tensors and plain containers (trainer.py:493-494 pickles both)."""
    import collections
    from ..config import defaults
    import dataclasses
    cfg_classes = [v for v in vars(defaults).values() if isinstance(v, type) and dataclasses.is_dataclass(v)]
    return [TrainingState, collections.OrderedDict, collections.defaultdict, dict] + cfg_classes


_HISTORY_KEYS = ("train_loss", "train_acc", "val_loss", "val_acc", "val_f1", "val_fpr", "val_fnr", "learning_rates")
_SAVE_EVERY = {"every_epoch": 1, "every_5_epochs": 5, "every_10_epochs": 10}


class Trainer:
    MAX_GRAPHS = 2

    def __init__(self, model: nn.Module, train_loader: DataLoader, val_loader: DataLoader, config: Any,
                 checkpoint_dir: Optional[Path] = None, device: str = "cuda", criterion: Optional[nn.Module] = None):
        enforce_cuda()
        self.device, self.config = device, config
        self.model = model.to(device).to(memory_format=torch.channels_last)
        self.native = isinstance(self.model, CNNSmallWakeword)
        self.train_loader, self.val_loader = train_loader, val_loader

        if criterion is None:
            criterion = create_loss_function(loss_name=config.loss.loss_function, num_classes=config.model.num_classes,
                                             label_smoothing=config.loss.label_smoothing,
                                             focal_alpha=config.loss.focal_alpha, focal_gamma=config.loss.focal_gamma,
                                             class_weights=None, device=device)
        self.criterion = criterion.to(device)
        self._native_loss = isinstance(self.criterion, _NativeLoss)
        if self._native_loss:
            self.criterion.validate_targets = False      # checked from the per-step stats read instead

        self.optimizer, self.scheduler = create_optimizer_and_scheduler(self.model, config)
        self.use_mixed_precision = config.optimizer.mixed_precision
        hip_backed = self.native or bool(getattr(self.model, "hip_backed", False))
        if self.native and self.use_mixed_precision:
            # the reference's AMP switch (fp16 autocast + GradScaler, trainer.py:92,172) maps to the HIP path's reduced-
            # precision modes: 16-bit activation/gradient storage, fp32 arithmetic.  amp_dtype "bf16" (default; fp32's
            # range, no loss scaling) or "fp16" (the reference's own type, with the device-side GradScaler below)
            amp = str(getattr(config.optimizer, "amp_dtype", "bf16")).lower()
            self.model.set_act_dtype(amp)
            logger.info("mixed_precision=True -> %s activation storage on the HIP path", amp)
        # HIP-backed models carry their precision mode themselves (act_dtype / mode at construction): never autocast
        self._autocast = self.use_mixed_precision and not hip_backed
        self.scaler = create_grad_scaler(enabled=self._autocast)
        self.loss_scale = None             # device ww_loss_scale when the model stores fp16 (set below, needs the optimizer)
        self.gradient_clip = config.optimizer.gradient_clip

        self.train_metrics_tracker = MetricsTracker(device=device)
        self.val_metrics_tracker = MetricsTracker(device=device)
        self.metric_monitor = MetricMonitor(window_size=100)
        self.state = TrainingState()               # the reference's fields, nothing added (checkpoint schema, trainer.py:33-41)
        # every LAUNCHED training batch, skipped or not: the counter of all Philox streams (SpecAugment, audio augmentation,
        # dropout), so a skipped batch never makes later batches reuse its masks; saved as the checkpoint's one extra key
        # so a resumed run continues the streams (state.global_step counts only batches that reported a result, :208)
        self.launched_steps = 0
        self.early_stopping_patience = config.training.early_stopping_patience
        self.checkpoint_dir = Path(checkpoint_dir) if checkpoint_dir is not None else Path("checkpoints")
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        self.checkpoint_frequency = config.training.checkpoint_frequency
        self.callbacks = []
        self.show_progress = sys.stderr.isatty()

        # data parallel: one process per GPU, gradients averaged over the flat bucket (RCCL via torch.distributed)
        import torch.distributed as dist
        self._dist = dist if (dist.is_available() and dist.is_initialized()
                              and getattr(config.training, "data_parallel", True)) else None
        # two gradient buckets, the late layers' reduced under the early layers' backward (_step_native: cnn_small's 82 KB bucket,
        # only when asked; _step_autograd_async: bucketed autograd models -- asked for, or by bucket size when "auto")
        _ov = getattr(config.training, "dp_overlap", False)
        self.dp_overlap = _ov is True
        self._dp_overlap_auto = isinstance(_ov, str) and _ov.lower() == "auto"
        self._ov = None                    # overlap state of a bucketed autograd model: cut offset, hooks, countdown
        self.last_collective = None        # which form the last data-parallel step used (bench / probes report it)
        self.world_size = self._dist.get_world_size() if self._dist else 1
        self.rank = self._dist.get_rank() if self._dist else 0
        if self._dist:
            with torch.no_grad():
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    self._dist.broadcast(t, src=0)
        self._stats_host = None
        # native step pipelining (see _launch_native): the optimizer consumes the device-side found_inf flag, so no
        # host read sits between backward and optimizer.step(); with deferred_metrics a step's 48-byte stats are
        # resolved while the NEXT step is already running (callbacks fire one step late, same order and content).
        self._fused_optimizer = isinstance(self.optimizer, FlatFusedOptimizer)    # clip + update = one launch
        if self._stores_fp16():
            # fp16 storage: gradients travel times a dynamic loss scale; trainer.scaler keeps GradScaler's interface
            if not (self._fused_optimizer and self._native_loss):
                raise nat.NativeError("fp16 storage needs the fused clip + optimizer step and the native loss (adam / adamw / "
                                      "sgd on a bucketed HIP model): the loss scale lives on the device between them")
            from .optimizer_factory import DeviceGradScaler
            self.scaler = DeviceGradScaler(self.device, self.optimizer)
            self.loss_scale = self.scaler.state
            self.criterion.loss_scale = self.loss_scale
        self._skip_on_device = self._fused_optimizer or bool(getattr(self.optimizer, "_step_supports_amp_scaling", False))
        # HIP-backed models that go through autograd (gru, crnn, mobilenetv3) get the same sync-free step: native loss stats,
        # device-side clip and skip decision, one deferred 48-byte read (the reference's step reads loss/acc/grad-norm back
        # three times per batch, trainer.py:177-209)
        self._async_autograd = (not self.native and self._native_loss and self._skip_on_device
                                and bool(getattr(self.model, "hip_backed", False)) and torch.device(device).type == "cuda")
        self.deferred_metrics = (bool(getattr(config.training, "deferred_metrics", True))
                                 and (self.native or self._async_autograd) and self._native_loss and self._skip_on_device)
        self._pending = None
        self._dropout_modules = [m for m in self.model.modules() if hasattr(m, "dropout_step")]
        self._dp_voted = False             # the ranks agreed to skip the current batch (reference-style step only)
        self._dp_pack = None               # (key, flat buffer, views) of the packed gradient all-reduce
        self._host_bufs, self._buf_i = None, 0
        self._in_stream = None
        # log-mel workgroups while the input stage runs BESIDE a step (side stream): one per CU.  The kernel is persistent;
        # a full-device grid (2-4 workgroups per CU) finishes sooner but takes registers / LDS from the conv kernels on every
        # CU (measured: 1.36 ms per step at full residency, 1.32 at one per CU); 0 = fill the device
        self.input_stage_workgroups = None
        # HIP graph replay of the native step (BASELINE config 5 "hipGraph-captured step"; see _graph_capture)
        # on when asked for, or unasked for a model that prefers it (replayed steps are bit-identical to eager ones, tests/test_hip_graph.py)
        # (the unasked form only where the in-graph gradient all-reduce can be captured: no process group, or RCCL -- gloo cannot)
        capturable = self._dist is None or str(self._dist.get_backend()).lower() == "nccl"
        self._graph_asked = bool(getattr(config.training, "hip_graph", False))     # asked for: a failing capture raises
        self.use_hip_graph = self._graph_asked or (
            bool(getattr(config.training, "hip_graph_auto", True)) and bool(getattr(model, "prefers_hip_graph", False)) and capturable)
        # captured graphs by _graph_key (feature shape + everything a capture bakes by value): the full batch and the ragged
        # last batch of an epoch each keep theirs (MAX_GRAPHS of them; further shapes run eagerly)
        self._graphs = {}
        self._graph = None                 # the graph used last: captured graph + static buffers + host mirror of the control block
        self._graph_seen = {}              # _graph_key -> eager steps seen (a key is captured when it repeats)
        self._eager_native_steps = 0
        self.audio_augmentation = None     # data.augmentation.AudioAugmentation; applied to (B,N) training batches
        logger.info("Trainer initialized (device=%s, model=%s, native=%s, optimizer=%s, scheduler=%s, loss=%s, "
                    "world=%d)", device, config.model.architecture, self.native, config.optimizer.optimizer,
                    config.optimizer.scheduler, config.loss.loss_function, self.world_size)

    # ------------------------------------------------------------------------------- helpers
    def _bar(self, loader, desc):
        if tqdm is None or not self.show_progress:
            return loader
        return tqdm(loader, desc=desc, leave=False)

    @staticmethod
    def _unpack(batch, idx, what):
        if not isinstance(batch, (tuple, list)) or len(batch) < 2:
            logger.error("Invalid batch structure at %s %d, skipping", what, idx)
            return None
        inputs, targets = batch[0], batch[1]
        if inputs.numel() == 0 or targets.numel() == 0:
            logger.warning("Empty tensor in %s %d, skipping", what, idx)
            return None
        return inputs, targets

    def _features(self, inputs: torch.Tensor, training: bool, step: Optional[int] = None) -> torch.Tensor:
        """Native front end for raw waveform batches (B,N): fused log-mel/MFCC (+SpecAugment when training)."""
        d, a = self.config.data, self.config.augmentation
        mfcc = d.feature_type == "mfcc"
        if d.feature_type not in ("mel", "mfcc", "mel_spectrogram"):
            raise ValueError(f"Unknown feature_type: {d.feature_type}")
        cfg = nat.make_feat_cfg(sample_rate=d.sample_rate, n_fft=d.n_fft, hop=d.hop_length, n_mels=d.n_mels,
                                n_mfcc=d.n_mfcc if mfcc else 0)
        sa = None
        if training and (a.n_freq_masks + a.n_time_masks) > 0 and (a.freq_mask_prob > 0 or a.time_mask_prob > 0):
            sa = nat.make_specaug_cfg(a.freq_mask_param, a.time_mask_param, a.n_freq_masks, a.n_time_masks,
                                      a.freq_mask_prob, a.time_mask_prob)
        wave = inputs.to(self.device, non_blocking=True)
        if wave.dtype not in (torch.float32, torch.int16):
            wave = wave.float()
        if training and self.audio_augmentation is not None:
            # RIR + background mix on the device, ahead of the STFT (the reference does this on CPU dataset workers)
            wave = self.audio_augmentation(wave, step=self._launch_step_index() if step is None else step,
                                           sample_offset=self.rank * wave.shape[0])
        return nat.logmel_fwd(wave.contiguous(), cfg, sa, seed=a.seed,
                              step=self._launch_step_index() if step is None else step,
                              sample_offset=self.rank * wave.shape[0])

    def _read_stats(self, stats: torch.Tensor) -> dict:
        """The step's single device->host transfer (48 bytes, pinned)."""
        if self._stats_host is None:
            self._stats_host = torch.empty(nat.STEP_STATS_BYTES, dtype=torch.uint8).pin_memory()
        self._stats_host.copy_(stats, non_blocking=True)
        torch.cuda.current_stream(stats.device).synchronize()
        return nat.decode_stats(self._stats_host)

    # ------------------------------------------------------------------------------- data parallel
    def _reduce_start(self, t: torch.Tensor):
        """Start averaging ``t`` over the ranks.  RCCL: asynchronous on the process group's stream -- it waits for what
        the current stream has queued so far and runs beside whatever is queued next (the rest of the backward)."""
        if self._dist.get_backend() == "nccl":
            return self._dist.all_reduce(t, op=self._dist.ReduceOp.AVG, async_op=True), None
        return self._dist.all_reduce(t, async_op=True), t           # gloo (CPU tests) has no AVG

    def _reduce_inline(self, t: torch.Tensor):
        """Average ``t`` over the ranks on the CURRENT stream (RCCL: no stream hand-off; ordered like any kernel)."""
        if self._dist.get_backend() == "nccl":
            self._dist.all_reduce(t, op=self._dist.ReduceOp.AVG)
        else:
            self._dist.all_reduce(t)
            t.div_(self.world_size)

    def _reduce_finish(self, handles):
        """The current stream (RCCL) / the host (gloo) waits for the started reductions."""
        for work, t in handles:
            work.wait()
            if t is not None:
                t.div_(self.world_size)

    def _pack_buffer(self):
        """Persistent flat fp32 buffer for the gradients of a model without buckets of its own, plus the found_inf slot."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        key = tuple(p.numel() for p in params) + (params[0].device, params[0].dtype)
        if self._dp_pack is None or self._dp_pack[0] != key:
            buf = torch.zeros(sum(key[:-2]) + 1, dtype=params[0].dtype, device=params[0].device)
            views, off = [], 0
            for p in params:
                views.append(buf[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._dp_pack = (key, buf, views)
        return self._dp_pack[1], self._dp_pack[2], params

    def _dp_flag_slot(self):
        """float32[1] device view that travels with the gradients through the all-reduce: any rank's found_inf."""
        if not self._dist or torch.device(self.device).type != "cuda":
            return None
        if hasattr(self.model, "flat_grad_ext"):
            if self.native:
                self.model._prepare(torch.device(self.device))
            return self.model.flat_grad_ext[-1:]
        return self._pack_buffer()[0][-1:]

    def _allreduce_grads(self):
        """Average the gradients over the ranks with ONE collective: a bucketed model's flat bucket as it is, any other
        model's gradients packed into a persistent flat buffer (two multi-tensor copies around the all-reduce instead of
        one latency-bound collective per parameter tensor -- 45 of them for the CRNN).  The buffer's last element is the
        found_inf slot (``_dp_flag_slot``), so the skip decision is reduced with the gradients."""
        if not self._dist:
            return
        if hasattr(self.model, "flat_grad_ext") and self.model.flat_grad_ext is not None and (
                self.native or not any(p.grad is None for p in self.model.parameters())):
            if not self.native:
                self.model.gather_grads()
            self._reduce_inline(self.model.flat_grad_ext)
            if not self.native:
                # the averaged values live in the bucket; a gradient that autograd allocated elsewhere (layers that do not
                # write into their grad_slot: the recurrent ones) still holds the LOCAL values -- rebind it to its bucket view,
                # so clip_gradients / torch.optim / a later gather_grads() all see the reduced gradient
                for v, p_ in zip(self.model._fb_views, self.model._fb_plist):
                    if p_.grad.data_ptr() != v.data_ptr():
                        p_.grad = v
            return
        buf, views, params = self._pack_buffer()
        have = [(v, p.grad) for v, p in zip(views, params) if p.grad is not None]
        if not have:
            return
        missing = [v for v, p in zip(views, params) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._reduce_inline(buf)
        torch._foreach_copy_([g for _, g in have], [v for v, _ in have])

    # Overlapped all-reduce for bucketed autograd models (MobileNetV3: one 5.8 MB fp32 bucket).  The bucket is in parameters()
    # order, so its TAIL holds the late layers, whose gradients the backward produces FIRST.  The tail (>= half of the bytes:
    # from the last inverted-residual blocks on) is all-reduced on the process group's stream as soon as its last gradient
    # has been accumulated, under the backward of the early layers; the head follows in-stream after the backward.  Each
    # hand-off to the collective's stream and back stalls the main queue ~20 us on this stack (profiles/
    # r02_allreduce_microbench.json), so the split only pays when the collective it hides is longer than that: "auto"
    # takes it from DP_OVERLAP_MIN_BYTES up (4 MB: ~40 us of hand-offs against >= 40 us of exposed all-reduce at the
    # ~100 GB/s a direct xGMI reduce-scatter + all-gather moves a bucket of this size; UNMEASURED at N > 1 on this box).
    DP_OVERLAP_MIN_BYTES = 4 << 20

    def _overlap_wanted(self) -> bool:
        if not (self._dist and self._fused_optimizer and hasattr(self.model, "flat_grad_ext") and not self.native):
            return False
        if self.dp_overlap:
            return True
        return self._dp_overlap_auto and self.model.flat_grad.numel() * 4 >= self.DP_OVERLAP_MIN_BYTES

    def _overlap_arm(self):
        """Hooks on the tail's parameters (once): the LAST of them to receive its gradient starts the tail's all-reduce."""
        plist, views = self.model._fb_plist, self.model._fb_views
        if self._ov is not None:
            if self._ov["ptr"] == self.model.flat_param.data_ptr():
                return self._ov
            for h in self._ov["hooks"]:                            # the bucket was rebuilt (model.to(...)): new slots, new hooks
                h.remove()
        total, acc, first = self.model.flat_grad.numel(), 0, len(plist)
        for i in range(len(plist) - 1, -1, -1):                    # smallest tail holding at least half of the elements
            acc += plist[i].numel()
            first = i
            if 2 * acc >= total:
                break
        ov = {"ptr": self.model.flat_param.data_ptr(), "first": first, "cut": total - acc, "left": 0, "handles": [],
              "n_tail": len(plist) - first, "hooks": [], "armed": False}

        ext_dev = self.model.flat_param.device

        def hook(param, _ov=ov, _self=self):
            if not _ov["armed"]:
                return
            _ov["left"] -= 1
            if _ov["left"] == 0:                                  # the tail is complete: gather what was born outside, reduce
                nat.deferred_flush(ext_dev)                       # partial sums the tail's backward nodes left queued
                ext = _self.model.flat_grad_ext
                with torch.no_grad():
                    todo = [(v, q) for v, q in zip(views[_ov["first"]:], plist[_ov["first"]:]) if q.grad.data_ptr() != v.data_ptr()]
                    if todo:
                        torch._foreach_copy_([v for v, _ in todo], [q.grad for _, q in todo])
                        for v, q in todo:
                            q.grad = v                            # gather_grads() after the backward must not copy them again
                _ov["handles"].append(_self._reduce_start(ext[_ov["cut"]:]))     # incl. the found_inf slot behind the bucket
        for q in plist[first:]:
            ov["hooks"].append(q.register_post_accumulate_grad_hook(hook))
        self._ov = ov
        return ov

    def _vote_skip(self, bad: bool) -> bool:
        """Reference-style step in data-parallel mode: the ranks agree BEFORE the gradient all-reduce whether this batch is
        skipped (a non-finite loss or an exception in the forward pass on one rank), so no rank is left waiting in a
        collective its peers never enter."""
        if not self._dist:
            return bad
        dev = self.device if torch.device(self.device).type == "cuda" else "cpu"
        flag = torch.tensor([1.0 if bad else 0.0], device=dev)
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MAX)
        return bool(flag.item() > 0)

    # ------------------------------------------------------------------------------- inner steps
    def _launch_step_index(self) -> int:
        """Counter of the Philox streams (SpecAugment, audio augmentation, dropout): the index of the step being
        launched -- every launched batch takes one, skipped or not."""
        return self.launched_steps

    def _begin_step(self, index: Optional[int] = None) -> int:
        """Hand the step's Philox counter to every HIP module that draws masks, then advance it."""
        if index is None:
            index = self.launched_steps
            self.launched_steps += 1
        for m in self._dropout_modules:
            m.dropout_step = index
        return index

    def _resolve(self, pending):
        """Read one launched step's stats (waits for that step only) -> (batch_idx, loss, acc) or None if skipped."""
        batch_idx, buf, event = pending
        event.synchronize()
        s = nat.decode_stats(buf)
        if s["bad_target"]:
            logger.error("Unexpected error at batch %d: Target values must be in [0, 1]", batch_idx)
            return None
        if s["found_inf"] != 0.0:
            if s["nonfinite"] or self.loss_scale is None:
                logger.error("Non-finite loss detected at batch %d: %s", batch_idx, s["loss"])
            else:      # GradScaler's skipped step: finite loss, gradients overflowed fp16 -- the scale has backed off
                logger.warning("Gradient overflow at batch %d (loss %.6f): step skipped, loss scale reduced", batch_idx, s["loss"])
            return None
        self.train_metrics_tracker.update_counts(s["tp"], s["tn"], s["fp"], s["fn"])
        self.last_grad_norm = s["grad_norm"]
        return batch_idx, s["loss"], s["correct"] / max(s["count"], 1)

    def _flush_pending(self):
        out = []
        if self._pending is not None:
            r = self._resolve(self._pending)
            self._pending = None
            if r is not None:
                out.append(r)
        return out

    def _prepare_native(self, inputs, targets, step_index):
        """Input stage of a native step on its own HIP stream: H2D (if needed) + fused log-mel/SpecAugment.  It has no
        dependency on the model, so for batch k+1 it runs while step k's conv stack is executing.  The overlap is not free:
        the resident log-mel workgroups take registers / LDS from the conv kernels, which is why the
        kernel is launched with ``input_stage_workgroups`` persistent workgroups here (one per CU) instead of a full-device
        grid.  Returns (features, targets, ready_event, step_index)."""
        if self._in_stream is None:
            self._in_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._in_stream):
            if inputs.dim() == 2:
                if self.input_stage_workgroups is None:
                    self.input_stage_workgroups = torch.cuda.get_device_properties(self.device).multi_processor_count
                nat.set_logmel_workgroups(self.device, self.input_stage_workgroups)
                try:
                    feats = self._features(inputs, training=True, step=step_index)
                finally:
                    nat.set_logmel_workgroups(self.device, 0)
            else:
                feats = inputs.to(self.device, non_blocking=True)
            tg = targets.to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._in_stream)
        return feats, tg, ready, step_index

    def _step_native(self, inputs, targets, batch_idx, prepared=None):
        """Launch one native step; returns the list of steps whose results became available."""
        if prepared is None:
            prepared = self._prepare_native(inputs, targets, self._begin_step())
        inputs, targets, ready, step_index = prepared
        self._begin_step(step_index)               # dropout draws from the same counter as the batch's SpecAugment
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ready)
        inputs.record_stream(main)
        targets.record_stream(main)
        replayed = self._graph_step_or_capture(inputs, targets, step_index, batch_idx)
        if replayed is not None:
            return replayed
        self.model.sample_offset = self.rank * inputs.shape[0]
        self.optimizer.zero_grad(set_to_none=True)
        flag = None
        if self.loss_scale is not None:
            self.criterion.loss_scale_slot = self.optimizer.scale_slot      # the loss kernel and the optimizer agree on it
        if not self._dist:
            stats = self.model.train_step_native(inputs, targets, self.criterion)     # fwd, loss, bwd: three C-ABI calls
        else:
            # The skip flag rides in the bucket's spare last float.  Default: ONE in-stream all-reduce of the whole bucket
            # after the backward -- on this stack a collective issued on the step's own stream costs its latency only
            # (2 us at one rank), while every hand-off to the collective's stream and back stalls the main queue ~20 us
            # (profiles/r02_allreduce_microbench.json), more than an 82 KB bucket can hide.  dp_overlap=True selects the
            # two-bucket form: blocks 2, 3 + classifier reduced on RCCL's stream under the backward of
            # blocks 1, 0 and the stem -- for buckets large enough to be worth two hand-offs.
            self.model._prepare(inputs.device)
            ext, cut, handles = self.model.flat_grad_ext, self.model.late_offset, []
            flag = ext[-1:]
            if self.dp_overlap and cut > 0:
                stats = self.model.train_step_native(inputs, targets, self.criterion, found_inf_out=flag,
                                                     mid_hook=lambda: handles.append(self._reduce_start(ext[cut:])))
                handles.append(self._reduce_start(ext[:cut]))
            else:
                stats = self.model.train_step_native(inputs, targets, self.criterion, found_inf_out=flag)
                self._reduce_inline(ext)
            self._reduce_finish(handles)
        if self._host_bufs is None:
            self._host_bufs = [torch.empty(nat.STEP_STATS_BYTES, dtype=torch.uint8).pin_memory() for _ in range(2)]
        # two pinned records, alternating: the fused optimizer's slot parity picks one (the same rule a graph replay follows)
        buf = self._host_bufs[self.optimizer._parity if self._fused_optimizer else self._buf_i]
        self._buf_i ^= 1
        if self._fused_optimizer:
            # clip_gradients + "skip a non-finite batch" + optimizer.step() (trainer.py:177-193) in ONE launch, which also
            # writes the step's 48-byte record into pinned host memory
            self.optimizer.step(max_norm=max(float(self.gradient_clip), 0.0), stats=stats, stats_host=buf,
                                found_inf_extra=flag, loss_scale=self.loss_scale)
        else:
            if flag is not None:                   # some rank's bad batch -> every rank's found_inf
                sv = stats.view(torch.float32)
                sv[nat.FOUND_INF_FLOAT_INDEX] = torch.maximum(sv[nat.FOUND_INF_FLOAT_INDEX], (flag[0] != 0).float())
            nat.grad_norm_clip_(self.model.flat_grad, max(float(self.gradient_clip), 0.0), stats=stats)
        if self._skip_on_device and not self._fused_optimizer:
            # the reference skips a batch whose loss is not finite (trainer.py:177-179) or whose targets are invalid
            # (losses.py:72) before touching the parameters; here the fused optimizer gets that decision as a
            # device flag (the mechanism GradScaler uses), so nothing on the host waits for the GPU
            self.optimizer.grad_scale = None
            self.optimizer.found_inf = stats.view(torch.float32)[nat.FOUND_INF_FLOAT_INDEX]
            try:
                self.optimizer.step()
            finally:
                del self.optimizer.grad_scale, self.optimizer.found_inf
        if not self._fused_optimizer:
            buf.copy_(stats, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        launched = (batch_idx, buf, event)
        self._eager_native_steps += 1
        self._graph_after_eager(inputs)
        if not self._skip_on_device:               # optimizer cannot skip on the device: decide on the host
            r = self._resolve(launched)
            if r is None:
                self.optimizer.zero_grad(set_to_none=True)
                return []
            self.optimizer.step()
            return [r]
        if not self.deferred_metrics:
            r = self._resolve(launched)
            return [] if r is None else [r]
        done = self._flush_pending()               # previous step: finished long ago, no stall
        self._pending = launched
        return done

    # ------------------------------------------------------------------------------- HIP graph replay of the native step
    def _graph_capture(self, feats):
        """Capture ONE training step for feature batches shaped like ``feats`` as a HIP graph:

            ww_step_ctl_advance   (step += 1, parity ^= 1 in the device control block)
            cnn_small forward -> loss -> backward [-> all-reduce] -> fused clip + optimizer  on feats_cur / tgt_cur,
            dropout's Philox step, the learning rate and the optimizer's step_state slot read from the control block

        Everything a step changes from one replay to the next lives in device memory (include/wwhip.h ww_step_ctl); the
        host copies the batch into the static buffers and replays.  The input stage (log-mel + SpecAugment of the NEXT
        batch) stays a host-issued launch on the side stream: a fork inside the graph does not run concurrently with
        the main branch on this stack (measured), the side stream does.  Replayed steps are bit-identical to
        eager ones (tests/test_hip_graph.py)."""
        dev = torch.device(self.device)
        B = feats.shape[0]
        g = {"in_shape": tuple(feats.shape), "act": self._graph_mode_key(), "key": self._graph_key(feats.shape)}
        g["feats_cur"] = torch.zeros(feats.shape, dtype=torch.float32, device=dev)
        g["tgt_cur"] = torch.zeros(B, dtype=torch.int64, device=dev)
        g["ctl"] = nat.step_ctl_new(dev, step=0, lr=get_learning_rate(self.optimizer), parity=0)
        g["step"], g["parity"], g["lr"] = 0, 0, get_learning_rate(self.optimizer)
        # host-side counters the captured Python code bumps (capturing runs nothing on the device): BatchNorm's pending
        # num_batches_tracked of the conv stack -- restored here, re-applied per replay
        counted = [m for m in self.model.modules() if hasattr(m, "_pending_tracked")]
        opt_parity, tracked = self.optimizer._parity, [m._pending_tracked for m in counted]
        # a fork inside a captured graph does not run beside the main branch on this stack (measured: CRNN B=512 7.59 ms
        # forked vs 7.15 ms serial vs 5.38 ms eager with the GRU directions on two streams), so capture them in line
        forked = [m for m in self.model.modules() if getattr(m, "overlap_directions", False)]
        for m in forked:
            m.overlap_directions = False
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        nat.bind_step_ctl(dev, g["ctl"])           # launches issued while bound read step / lr / parity from the block
        max_norm = max(float(self.gradient_clip), 0.0)
        try:
            with torch.cuda.graph(graph):
                nat.step_ctl_advance(dev)
                for m in self._dropout_modules:
                    m.dropout_step = 0             # an OFFSET while the block is bound
                if hasattr(self.model, "sample_offset"):
                    self.model.sample_offset = self.rank * B
                self.optimizer.zero_grad(set_to_none=True)
                flag = self.model.flat_grad_ext[-1:] if self._dist else None
                if self.native:
                    stats = self.model.train_step_native(g["feats_cur"], g["tgt_cur"], self.criterion, found_inf_out=flag)
                else:                              # HIP-backed autograd model (crnn, gru, mobilenetv3) with flat buckets
                    self.criterion.found_inf_out = flag
                    self.criterion(self.model(g["feats_cur"]), g["tgt_cur"]).backward()
                    stats = self.criterion.last_stats
                    self.model.gather_grads()
                if self._dist:
                    self._reduce_inline(self.model.flat_grad_ext)
                self.optimizer.step(max_norm=max_norm, stats=stats, stats_host=self._host_bufs[0],
                                    stats_host_alt=self._host_bufs[1], found_inf_extra=flag, gathered=not self.native,
                                    loss_scale=self.loss_scale)
        finally:
            nat.bind_step_ctl(dev, None)
            for m in forked:
                m.overlap_directions = True
            self.optimizer._parity = opt_parity    # capturing ran nothing
            g["tracked"] = [(m, m._pending_tracked - t0) for m, t0 in zip(counted, tracked)]
            for m, t0 in zip(counted, tracked):
                m._pending_tracked = t0
            if self.native:
                for p_ in self.model._plist:       # .grad stay the views of the flat bucket every replay writes
                    p_.grad = self.model._grad_views[id(p_)]
        g["graph"] = graph
        # buffers the graph's nodes point at but that were allocated outside its memory pool: keep them alive with it
        g["keepalive"] = [self.model.flat_grad_ext, self.model.flat_param, self.criterion.last_stats] + self._host_bufs
        g["keepalive"] += [slot["buf"] for m in self.model.modules() if hasattr(m, "_ws") for slot in m._ws.values()]
        self._graphs[g["key"]] = g
        self._graph = g
        logger.info("HIP graph captured for the native step (features %s)", g["in_shape"])

    def _step_native_graph(self, feats, targets, idx, batch_idx):
        """Replay the captured step on this batch (its input stage has been waited for on the current stream)."""
        g = self._graph
        if self.model.flat_param.data_ptr() != self.optimizer._flat_ptr:      # the check an eager optimizer.step() makes
            raise nat.NativeError("the model's parameter bucket moved after the optimizer was created "
                                  "(model.to(...) / dtype change): create the optimizer afterwards")
        g["feats_cur"].copy_(feats)
        g["tgt_cur"].copy_(targets)
        # host mirror of the control block: rewritten only when it would be wrong (lr change, eager step in between)
        lr = get_learning_rate(self.optimizer)
        before = ((idx - 1) & 0xFFFFFFFFFFFFFFFF, self.optimizer._parity ^ 1)
        if (g["step"], g["parity"]) != before:
            nat.step_ctl_write(g["ctl"], step=before[0], parity=before[1])
        if g["lr"] != lr:
            nat.step_ctl_write(g["ctl"], lr=lr)
            g["lr"] = lr
        g["graph"].replay()
        g["step"], g["parity"] = idx & 0xFFFFFFFFFFFFFFFF, self.optimizer._parity
        buf = self._host_bufs[self.optimizer._parity]
        self.optimizer._parity ^= 1
        for m, d in g["tracked"]:
            m._pending_tracked += d
        event = torch.cuda.Event()
        event.record()
        launched = (batch_idx, buf, event)
        if not self.deferred_metrics:
            r = self._resolve(launched)
            return [] if r is None else [r]
        done = self._flush_pending()
        self._pending = launched
        return done

    def _stores_fp16(self) -> bool:
        # (`act` is the storage code of the conv stack only -- the generic layers of MobileNetV3 use that name for their
        # activation function)
        return any((isinstance(m, CNNSmallWakeword) and m.act == nat.ACT_F16) or getattr(m, "mode", None) is torch.float16
                   for m in self.model.modules())

    def _graph_mode_key(self):
        """What a captured graph bakes besides shapes: the storage / matrix modes of the model's HIP modules."""
        return tuple((m.act if isinstance(m, CNNSmallWakeword) else None, str(getattr(m, "mode", None)))
                     for m in self.model.modules() if isinstance(m, CNNSmallWakeword) or hasattr(m, "mode"))

    def _graph_key(self, shape):
        """Everything a captured step bakes BY VALUE: feature shape, storage / matrix modes, the clip norm, the optimizer's
        hyper-parameters other than lr (lr, the Philox step and the slot parity live in the device control block), the loss
        hyper-parameters, and where the parameter bucket is.  A change of any of them makes the next step eager and the one
        after it a fresh capture, exactly as an eager step would honour the change."""
        def plain(d, skip=()):
            return tuple(sorted((k, tuple(v) if isinstance(v, (tuple, list)) else v) for k, v in d.items()
                                if k not in skip and isinstance(v, (int, float, bool, str, tuple, list, type(None)))))
        groups = tuple(plain(g, skip=("params", "lr", "initial_lr")) for g in self.optimizer.param_groups)
        crit = plain({k: v for k, v in vars(self.criterion).items() if k in ("_eps", "_alpha", "_gamma", "smoothing", "alpha", "gamma")})
        flat = getattr(self.model, "flat_param", None)
        return (tuple(shape), self._graph_mode_key(), float(self.gradient_clip), groups, type(self.criterion).__name__, crit,
                None if flat is None else flat.data_ptr())

    def _graph_step_or_capture(self, feats, targets, step_index, batch_idx):
        """Graph mode of a sync-free step: replay when a graph for this key (shape + baked values) exists, else None (the
        caller runs the eager step; ``_graph_after_eager`` then captures a key the second time it is seen)."""
        if not (self.use_hip_graph and self._graphs):
            return None
        g = self._graphs.get(self._graph_key(feats.shape))
        if g is None:
            return None
        self._graph = g
        return self._step_native_graph(feats, targets, step_index, batch_idx)

    def _graph_after_eager(self, feats):
        if not (self.use_hip_graph and self._fused_optimizer):
            return
        key = self._graph_key(feats.shape)
        if key in self._graphs:
            return
        seen = self._graph_seen.get(key, 0)
        self._graph_seen[key] = seen + 1
        if seen == 0:
            return                                 # second batch with this key: workspaces, tables and buckets are warm
        # graphs whose baked values went stale (same shape, other clip norm / hyper-parameters / bucket) are dropped first
        for k in [k for k in self._graphs if k[0] == key[0]]:
            del self._graphs[k]
        if len(self._graphs) >= self.MAX_GRAPHS:
            return                                 # the frequent shapes have theirs; this one stays eager
        try:
            self._graph_capture(feats)
        except Exception as e:                     # noqa: BLE001
            if self._graph_asked:
                raise                              # hip_graph=True was asked for: its failure is the caller's to see
            # an UNASKED capture (hip_graph_auto) must never end a run the eager step completes: log once, stay eager.
            # (_graph_capture's finally block has restored the optimizer parity, the BatchNorm counters, the control-block
            # binding and the gradient views; the step that preceded the attempt ran eagerly and is complete.)
            logger.warning("HIP graph capture failed (%s: %s); continuing with eager steps", type(e).__name__, e)
            self.use_hip_graph = False
            self._graphs.clear()
            self._graph = None
            self.optimizer.zero_grad(set_to_none=True)     # gradients of the aborted capture may live in its dropped pool
            torch.cuda.synchronize(self.device)

    def _step_autograd_async(self, inputs, targets, batch_idx):
        """Training step of a HIP-backed autograd model without host reads: forward / native loss / backward through
        autograd, gradient all-reduce, clip_grad_norm_ and the "skip a non-finite batch" decision on the device (the
        fused optimizer takes it as found_inf), statistics resolved one step later."""
        step_index = self._begin_step()
        inputs = self._to_model_input(inputs, training=True, step=step_index)
        targets = targets.to(self.device, non_blocking=True)
        if inputs.dim() == 4 and inputs.is_cuda and self._fused_optimizer:
            replayed = self._graph_step_or_capture(inputs.contiguous(), targets, step_index, batch_idx)
            if replayed is not None:
                return replayed
        if hasattr(self.model, "sample_offset"):
            self.model.sample_offset = self.rank * inputs.shape[0]
        self.optimizer.zero_grad(set_to_none=True)
        flag = self._dp_flag_slot()                # the loss kernel drops its skip flag behind the gradient bucket
        self.criterion.found_inf_out = flag
        if self.loss_scale is not None:
            self.criterion.loss_scale_slot = self.optimizer.scale_slot
        ov = self._overlap_arm() if self._overlap_wanted() else None
        if ov is not None:
            ov["left"], ov["handles"], ov["armed"] = ov["n_tail"], [], True
        try:
            self.criterion(self.model(inputs), targets).backward()     # no reference to the autograd graph survives this line: a
        finally:                                                       # later graph capture needs fresh AccumulateGrad nodes
            if ov is not None:
                ov["armed"] = False
        stats = self.criterion.last_stats
        if self._host_bufs is None:
            self._host_bufs = [torch.empty(nat.STEP_STATS_BYTES, dtype=torch.uint8).pin_memory() for _ in range(2)]
        # two pinned records, alternating: the fused optimizer's slot parity picks one (the same rule a graph replay follows)
        buf = self._host_bufs[self.optimizer._parity if self._fused_optimizer else self._buf_i]
        self._buf_i ^= 1
        if self._fused_optimizer:
            # flat buckets (models/flat_buckets.py): gather the autograd gradients, ONE all-reduce, then clip + skip +
            # update + the 48-byte statistics record in one launch -- as the native cnn_small step does
            self.model.gather_grads()
            if self._dist:
                if ov is not None and ov["handles"]:
                    self._reduce_inline(self.model.flat_grad_ext[:ov["cut"]])      # the head, in-stream
                    self._reduce_finish(ov["handles"])                             # the tail has been running since mid-backward
                    self.last_collective = "overlapped"
                else:
                    self._reduce_inline(self.model.flat_grad_ext)
                    self.last_collective = "in-stream"
            self.optimizer.step(max_norm=max(float(self.gradient_clip), 0.0), stats=stats, stats_host=buf, gathered=True,
                                found_inf_extra=flag, loss_scale=self.loss_scale)
        else:
            self._allreduce_grads()
            sv = stats.view(torch.float32)
            if flag is not None:
                sv[nat.FOUND_INF_FLOAT_INDEX] = torch.maximum(sv[nat.FOUND_INF_FLOAT_INDEX], (flag[0] != 0).float())
            if self.gradient_clip > 0:
                norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), float(self.gradient_clip), foreach=True)
                sv[1] = norm                                                               # ww_step_stats.grad_norm
                sv[nat.FOUND_INF_FLOAT_INDEX] = torch.maximum(sv[nat.FOUND_INF_FLOAT_INDEX],
                                                              (~torch.isfinite(norm)).float())
            self.optimizer.grad_scale = None
            self.optimizer.found_inf = sv[nat.FOUND_INF_FLOAT_INDEX]
            try:
                self.optimizer.step()
            finally:
                del self.optimizer.grad_scale, self.optimizer.found_inf
            buf.copy_(stats, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        launched = (batch_idx, buf, event)
        if inputs.dim() == 4 and inputs.is_cuda:
            self._graph_after_eager(inputs)
        if not self.deferred_metrics:
            r = self._resolve(launched)
            return [] if r is None else [r]
        done = self._flush_pending()
        self._pending = launched
        return done

    def _to_model_input(self, inputs, training, step=None):
        """(B,N) waveforms go through the native front end (log-mel/MFCC [+SpecAugment, audio augmentation]); feature
        batches are moved as the reference does (channels_last, trainer.py:160)."""
        if inputs.dim() == 2 and torch.device(self.device).type == "cuda":
            return self._features(inputs, training=training, step=step)
        return inputs.to(self.device, non_blocking=True, memory_format=torch.channels_last)

    def _step_generic(self, inputs, targets, batch_idx):
        step_index = self._begin_step()
        self.optimizer.zero_grad(set_to_none=True)
        dev_type = torch.device(self.device).type
        failure = None
        try:
            inputs = self._to_model_input(inputs, training=True, step=step_index)
            targets = targets.to(self.device, non_blocking=True)
            if hasattr(self.model, "sample_offset"):
                self.model.sample_offset = self.rank * inputs.shape[0]
            with torch.autocast(dev_type, enabled=self._autocast and dev_type == "cuda"):
                outputs = self.model(inputs)
                loss = self.criterion(outputs, targets)
            finite = bool(torch.isfinite(loss))
        except Exception as e:                     # noqa: BLE001 -- data parallel: the peers must learn of it before the all-reduce
            if not self._dist:
                raise
            failure, finite = e, False
        if self._vote_skip(not finite):            # all ranks skip together (no rank waits in a collective alone)
            if failure is not None:
                self._dp_voted = True              # the peers skip this batch too: the epoch loop may log + continue
                raise failure
            if not finite:
                logger.error("Non-finite loss detected at batch %d: %s", batch_idx, loss.item())
            else:
                logger.error("Batch %d skipped: another rank reported a non-finite loss or an error", batch_idx)
            return []
        self.scaler.scale(loss).backward()
        self._allreduce_grads()
        if self.gradient_clip > 0:
            self.scaler.unscale_(self.optimizer)
            self.last_grad_norm = clip_gradients(self.model, self.gradient_clip)
        else:
            self.last_grad_norm = 0.0
        self.scaler.step(self.optimizer)
        self.scaler.update()
        with torch.no_grad():
            acc = (outputs.argmax(dim=1) == targets).float().mean().item()
        self.train_metrics_tracker.update(outputs.detach(), targets.detach())
        return [(batch_idx, loss.item(), acc)]

    def _eval_batch(self, inputs, targets) -> Optional[float]:
        if self.native and self._native_loss:
            inputs = self._features(inputs, training=False) if inputs.dim() == 2 else inputs.to(self.device,
                                                                                              non_blocking=True)
            targets = targets.to(self.device, non_blocking=True)
            self.model.sample_offset = self.rank * inputs.shape[0]
            self.criterion(self.model(inputs), targets)
            s = self._read_stats(self.criterion.last_stats)
            if s["bad_target"]:
                raise ValueError("Target values must be in [0, 1]")
            if s["nonfinite"]:
                return None
            self.val_metrics_tracker.update_counts(s["tp"], s["tn"], s["fp"], s["fn"])
            return s["loss"]
        inputs = self._to_model_input(inputs, training=False)
        targets = targets.to(self.device, non_blocking=True)
        dev_type = torch.device(self.device).type
        with torch.autocast(dev_type, enabled=self._autocast and dev_type == "cuda"):
            outputs = self.model(inputs)
            loss = self.criterion(outputs, targets)
        if not torch.isfinite(loss):
            return None
        self.val_metrics_tracker.update(outputs.detach(), targets.detach())
        return loss.item()

    def _epoch_reduce(self, tracker: MetricsTracker, loss_sum: float, n_batches: int):
        """Data parallel: every rank sees the epoch's global counters / mean loss."""
        if not self._dist:
            return loss_sum, n_batches
        dev = self.device if torch.device(self.device).type == "cuda" else "cpu"
        t = torch.tensor(tracker._c + [n_batches], dtype=torch.float64, device=dev)
        ls = torch.tensor([loss_sum], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t)
        self._dist.all_reduce(ls)
        tracker._c = [int(v) for v in t[:4].tolist()]
        tracker._seen = tracker._seen or sum(tracker._c) > 0
        return float(ls.item()), int(t[4].item())

    # ------------------------------------------------------------------------------- epochs
    def train_epoch(self, epoch: int) -> Tuple[float, float]:
        self.model.train()
        self.train_metrics_tracker.reset()
        num_batches = len(self.train_loader)
        if num_batches == 0:
            logger.warning("Training loader is empty, skipping epoch")
            return 0.0, 0.0
        epoch_loss = 0.0
        step = (self._step_native if (self.native and self._native_loss)
                else self._step_autograd_async if self._async_autograd else self._step_generic)
        bar = self._bar(self.train_loader, f"Epoch {epoch + 1}/{self.config.training.epochs} [Train]")

        def account(done):
            nonlocal epoch_loss
            for idx, loss_value, batch_acc in done:
                self.metric_monitor.update_batch(loss_value, batch_acc)
                epoch_loss += loss_value
                if bar is not self.train_loader:
                    avg = self.metric_monitor.get_running_averages()
                    bar.set_postfix({"loss": f"{avg['loss']:.4f}", "acc": f"{avg['accuracy']:.4f}",
                                     "lr": f"{get_learning_rate(self.optimizer):.6f}"})
                self.state.global_step += 1
                self._call_callbacks("on_batch_end", idx, loss_value, batch_acc)

        pipelined = self.native and self._native_loss

        def survivable(e) -> bool:
            """The reference logs and skips a failed batch (trainer.py:214-226).  Data parallel, that is only safe when
            the peers skip it too -- i.e. the ranks voted (reference-style step); a rank that dropped out of a sync-free
            step would leave the others inside the gradient all-reduce, so there the error ends the run on every rank."""
            if not self._dist:
                return True
            voted, self._dp_voted = self._dp_voted, False
            return voted

        def stage(item):
            """fetch-side work for one loader item: validate, and (native) enqueue its input stage on the side stream"""
            idx, batch = item
            try:
                parsed = self._unpack(batch, idx, "batch")
                if parsed is None:
                    return None
                prep = None
                if pipelined:
                    index = self.launched_steps
                    self.launched_steps += 1          # the batch owns this Philox step whether or not it survives
                    prep = self._prepare_native(parsed[0], parsed[1], index)
                return idx, parsed, prep
            except RuntimeError as e:
                if "out of memory" in str(e).lower() and survivable(e):
                    logger.error("GPU OOM at batch %d. Clearing cache and skipping batch.", idx)
                    torch.cuda.empty_cache()
                    return None
                logger.exception("Runtime error at batch %d: %s", idx, e)
                raise
            except Exception as e:
                logger.exception("Unexpected error at batch %d: %s", idx, e)
                if not survivable(e):
                    raise
                return None

        it = iter(enumerate(bar))
        nxt = next(it, None)
        staged = None
        batch_idx = -1
        while nxt is not None or staged is not None:
            try:
                if staged is None and nxt is not None:      # prime / refill
                    staged, nxt = stage(nxt), next(it, None)
                    if staged is None:
                        continue
                cur = staged
                # look one batch ahead: its input stage overlaps the step launched below
                staged = None
                while nxt is not None and staged is None:
                    staged, nxt = stage(nxt), next(it, None)
                batch_idx, parsed, prep = cur
                if pipelined:
                    account(self._step_native(None, None, batch_idx, prepared=prep))
                else:
                    account(step(parsed[0], parsed[1], batch_idx))
            except RuntimeError as e:
                if "out of memory" in str(e).lower() and survivable(e):
                    logger.error("GPU OOM at batch %d. Clearing cache and skipping batch.", batch_idx)
                    torch.cuda.empty_cache()
                    continue
                logger.exception("Runtime error at batch %d: %s", batch_idx, e)
                raise
            except Exception as e:
                logger.exception("Unexpected error at batch %d: %s", batch_idx, e)
                if not survivable(e):
                    raise
                continue
        account(self._flush_pending())            # the last step's deferred results
        epoch_loss, num_batches = self._epoch_reduce(self.train_metrics_tracker, epoch_loss, num_batches)
        avg_loss = epoch_loss / max(num_batches, 1)
        train_metrics = self.train_metrics_tracker.compute()
        logger.info("Epoch %d [Train]: Loss=%.4f, %s", epoch + 1, avg_loss, train_metrics)
        return avg_loss, train_metrics.accuracy

    def validate_epoch(self, epoch: int) -> Tuple[float, MetricResults]:
        self.model.eval()
        self.val_metrics_tracker.reset()
        num_batches = len(self.val_loader)
        if num_batches == 0:
            logger.warning("Validation loader is empty")
            return 0.0, MetricResults.empty()
        epoch_loss = 0.0
        bar = self._bar(self.val_loader, f"Epoch {epoch + 1}/{self.config.training.epochs} [Val]")
        with torch.no_grad():
            for batch_idx, batch in enumerate(bar):
                try:
                    parsed = self._unpack(batch, batch_idx, "validation batch")
                    if parsed is None:
                        continue
                    loss_value = self._eval_batch(parsed[0], parsed[1])
                    if loss_value is None:
                        logger.warning("Non-finite validation loss at batch %d", batch_idx)
                        continue
                    epoch_loss += loss_value
                except RuntimeError as e:
                    if "out of memory" in str(e).lower():
                        logger.error("GPU OOM during validation at batch %d", batch_idx)
                        torch.cuda.empty_cache()
                        continue
                    logger.exception("Runtime error during validation at batch %d: %s", batch_idx, e)
                    raise
                except Exception as e:
                    logger.exception("Unexpected error during validation at batch %d: %s", batch_idx, e)
                    continue
        epoch_loss, num_batches = self._epoch_reduce(self.val_metrics_tracker, epoch_loss, num_batches)
        avg_loss = epoch_loss / max(num_batches, 1)
        val_metrics = self.val_metrics_tracker.compute()
        logger.info("Epoch %d [Val]: Loss=%.4f, %s", epoch + 1, avg_loss, val_metrics)
        return avg_loss, val_metrics

    # ------------------------------------------------------------------------------- full loop
    def train(self, start_epoch: int = 0, resume_from: Optional[Path] = None) -> Dict[str, Any]:
        if resume_from is not None:
            self.load_checkpoint(Path(resume_from))
            start_epoch = self.state.epoch + 1
            logger.info("Resumed from checkpoint at epoch %d", start_epoch)
        history = {k: [] for k in _HISTORY_KEYS}
        epochs = self.config.training.epochs
        logger.info("Starting training: epochs=%d, train batches=%d, val batches=%d, batch size=%d", epochs,
                    len(self.train_loader), len(self.val_loader), self.config.training.batch_size)
        t0 = time.time()
        try:
            for epoch in range(start_epoch, epochs):
                self.state.epoch = epoch
                self._call_callbacks("on_epoch_start", epoch)
                train_loss, train_acc = self.train_epoch(epoch)
                val_loss, vm = self.validate_epoch(epoch)
                self._update_scheduler(val_loss)
                lr = get_learning_rate(self.optimizer)
                for key, value in zip(_HISTORY_KEYS, (train_loss, train_acc, val_loss, vm.accuracy, vm.f1_score,
                                                      vm.fpr, vm.fnr, lr)):
                    history[key].append(value)
                self.val_metrics_tracker.save_epoch_metrics(vm)
                improved = self._check_improvement(val_loss, vm.f1_score, vm.fpr)
                self._save_checkpoint(epoch, val_loss, vm, improved)
                if self._should_stop_early():
                    logger.info("Early stopping triggered after %d epochs", epoch + 1)
                    break
                self._call_callbacks("on_epoch_end", epoch, train_loss, val_loss, vm)
                if self.rank == 0:
                    print(f"\nEpoch {epoch + 1}/{epochs}\n  Train: Loss={train_loss:.4f}, Acc={train_acc:.4f}\n"
                          f"  Val:   Loss={val_loss:.4f}, Acc={vm.accuracy:.4f}, F1={vm.f1_score:.4f}, "
                          f"FPR={vm.fpr:.4f}, FNR={vm.fnr:.4f}\n  LR: {lr:.6f}"
                          + ("\n  New best model (improvement detected)\n" if improved else ""))
        except KeyboardInterrupt:
            logger.info("Training interrupted by user")
        self.state.training_time = time.time() - t0
        best_f1_epoch, _ = self.val_metrics_tracker.get_best_epoch("f1_score")
        best_fpr_epoch, _ = self.val_metrics_tracker.get_best_epoch("fpr")
        logger.info("Training complete: %.2f h, best val loss %.4f, best F1 %.4f, best FPR %.4f",
                    self.state.training_time / 3600, self.state.best_val_loss, self.state.best_val_f1,
                    self.state.best_val_fpr)
        return {"history": history, "final_epoch": self.state.epoch, "best_val_loss": self.state.best_val_loss,
                "best_val_f1": self.state.best_val_f1, "best_val_fpr": self.state.best_val_fpr,
                "training_time": self.state.training_time, "best_f1_epoch": best_f1_epoch,
                "best_fpr_epoch": best_fpr_epoch}

    def _update_scheduler(self, val_loss: float):
        if self.scheduler is not None and hasattr(self.scheduler, "step"):
            try:
                self.scheduler.step(val_loss)      # positional, as the reference does (quirk Q2)
            except TypeError:
                self.scheduler.step()

    def _check_improvement(self, val_loss: float, val_f1: float, val_fpr: float) -> bool:
        s, improved = self.state, False
        if val_loss < s.best_val_loss:
            s.best_val_loss, improved = val_loss, True
        if val_f1 > s.best_val_f1:
            s.best_val_f1, improved = val_f1, True
        if val_fpr < s.best_val_fpr:
            s.best_val_fpr, improved = val_fpr, True
        s.epochs_without_improvement = 0 if improved else s.epochs_without_improvement + 1
        return improved

    def _should_stop_early(self) -> bool:
        return self.state.epochs_without_improvement >= self.early_stopping_patience

    # ------------------------------------------------------------------------------- checkpoints
    def _atomic_save(self, obj, path: Path, what: str):
        tmp = path.with_suffix(".pt.tmp")
        try:
            torch.save(obj, tmp)
            tmp.replace(path)
            logger.info("Saved %s: %s", what, path)
        except Exception as e:
            logger.error("Failed to save %s: %s", what, e)
            if tmp.exists():
                tmp.unlink()

    def _save_checkpoint(self, epoch: int, val_loss: float, val_metrics: MetricResults, is_best: bool):
        every = _SAVE_EVERY.get(self.checkpoint_frequency)
        periodic = (every is not None and (epoch + 1) % every == 0) or \
                   (self.checkpoint_frequency == "best_only" and is_best)
        if not (periodic or is_best) or self.rank != 0:
            return
        try:
            ckpt = {"epoch": epoch, "model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict() if self.scheduler else None,
                    "scaler_state_dict": self.scaler.state_dict(), "state": self.state, "config": self.config,
                    "val_loss": val_loss, "val_metrics": val_metrics.to_dict(),
                    "launched_steps": int(self.launched_steps)}      # extension: not in the reference's dict (:487-497)
        except Exception as e:
            logger.error("Failed to create checkpoint dict: %s", e)
            return
        if periodic:
            self._atomic_save(ckpt, self.checkpoint_dir / f"checkpoint_epoch_{epoch + 1:03d}.pt", "checkpoint")
        if is_best:
            self._atomic_save(ckpt, self.checkpoint_dir / "best_model.pt", "best model")

    def load_checkpoint(self, checkpoint_path: Path):
        checkpoint_path = Path(checkpoint_path)
        if not checkpoint_path.exists():
            raise FileNotFoundError(f"Checkpoint file not found: {checkpoint_path}")
        if not checkpoint_path.is_file():
            raise ValueError(f"Checkpoint path is not a file: {checkpoint_path}")
        try:
            # the dict pickles TrainingState / config instances, as the reference's does (:493-494); they are the ONLY
            # non-tensor classes the restricted unpickler accepts -- a checkpoint cannot run code in the training process
            with torch.serialization.safe_globals(_checkpoint_safe_globals()):
                ckpt = torch.load(checkpoint_path, map_location=self.device, weights_only=True)
        except Exception as e:
            raise RuntimeError(f"Corrupted or invalid checkpoint file (loaded with weights_only=True; classes other than "
                               f"TrainingState and the config dataclasses are refused): {checkpoint_path}: {e}") from e
        missing = [k for k in ("model_state_dict", "optimizer_state_dict", "state") if k not in ckpt]
        if missing:
            raise ValueError(f"Checkpoint missing required keys: {missing}")
        try:
            self.model.load_state_dict(ckpt["model_state_dict"])
        except Exception as e:
            raise RuntimeError("Model state dict incompatible with current model") from e
        for name, obj, key in (("optimizer", self.optimizer, "optimizer_state_dict"),
                               ("scheduler", self.scheduler, "scheduler_state_dict"),
                               ("scaler", self.scaler, "scaler_state_dict")):
            if obj is None or not ckpt.get(key):
                continue
            try:
                obj.load_state_dict(ckpt[key])
            except Exception as e:
                logger.warning("Failed to load %s state dict: %s. Continuing with fresh %s.", name, e, name)
        self.state = ckpt["state"]
        # a reference checkpoint has no launched-step counter: without skipped batches it equals global_step
        self.launched_steps = int(ckpt.get("launched_steps", self.state.global_step))
        # the Philox streams (SpecAugment, audio augmentation, dropout) continue where the interrupted run stood: all of
        # them are driven by self.launched_steps (_begin_step)
        for m in self._dropout_modules:
            m.dropout_step = int(self.launched_steps)
        logger.info("Checkpoint loaded: Epoch %d", self.state.epoch + 1)

    # ------------------------------------------------------------------------------- callbacks
    def add_callback(self, callback: Callable):
        self.callbacks.append(callback)

    def _call_callbacks(self, event: str, *args, **kwargs):
        for cb in self.callbacks:
            fn = getattr(cb, event, None)
            if fn is not None:
                fn(*args, **kwargs)
