"""Model factory of the hot path: ``create_model`` keeps the reference signature
(``src/models/architectures.py:437-509``) and adds ``cnn_small`` whose
forward/backward run entirely in libwwhip (hand-written HIP, gfx950).

The module holds ordinary ``nn.Parameter``s / buffers under the same names as the plain
``torch.nn`` formulation (``oracle/cnn_small.py``), so ``state_dict()``, ``load_state_dict()``,
``.to()``, ``.parameters()``, optimizers and checkpoints behave as for any reference model
(needed by ``src/training/trainer.py:69-71,489,551``).  There is no PyTorch fallback: off an
MI355X, or without the built library, ``forward`` raises.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _native as nat

_SUPPORTED = "cnn_small, gru, crnn, mobilenetv3"
_REFERENCE_ONLY = ("resnet18", "lstm", "tcn")


class _DSBlock(nn.Module):
    """Parameter container of one depthwise-separable block (never called)."""

    def __init__(self, ch):
        super().__init__()
        self.dw = nn.Conv2d(ch, ch, 3, padding=1, groups=ch, bias=False)
        self.dw_bn = nn.BatchNorm2d(ch)
        self.pw = nn.Conv2d(ch, ch, 1, bias=False)
        self.pw_bn = nn.BatchNorm2d(ch)


class _CNNSmallFn(torch.autograd.Function):
    """logits = cnn_small(x).  Parameter gradients are written by the HIP backward straight into the
    module's flat gradient bucket (``p.grad`` are views of it), so nothing is returned for them."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        logits, ws, step = mod._launch_forward(x, training=True)
        ctx.mod, ctx.ws, ctx.step = mod, ws, step
        ctx.save_for_backward(x)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        (x,) = ctx.saved_tensors
        ctx.mod._launch_backward(x, dlogits, ctx.ws, ctx.step)
        return (None, None) + (None,) * len(ctx.mod._plist)


class _CNNFrontFn(torch.autograd.Function):
    """seq = mean over frequency of the conv stack's last activations (the CRNN front-end)."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        seq, ws = mod._launch_front_forward(x, training=True)
        ctx.mod, ctx.ws = mod, ws
        ctx.save_for_backward(x)
        return seq

    @staticmethod
    def backward(ctx, dseq):
        (x,) = ctx.saved_tensors
        ctx.mod._launch_front_backward(x, dseq, ctx.ws)
        return (None, None) + (None,) * len(ctx.mod._plist)


class CNNSmallWakeword(nn.Module):
    """stem Conv2d(1,64,3,s2,p1)+BN+ReLU -> 4 x [DW3x3+BN+ReLU, PW1x1+BN+ReLU] -> GAP -> dropout -> Linear(64,2).

    ``forward(x)``: x (B,1,F,T) float32 on an MI355X -> (B,2) logits.
    """

    CH = 64
    N_BLOCKS = 4

    def __init__(self, num_classes: int = 2, pretrained: bool = False, dropout: float = 0.3,
                 input_channels: int = 1, dropout_seed: int = 0, act_dtype: str = "fp32", features_only: bool = False):
        super().__init__()
        self.features_only = bool(features_only)      # conv stack + frequency pooling only (the CRNN's front-end)
        self.act = nat.act_code(act_dtype)   # storage of the conv-stack activations: fp32 (parity) | bf16
        if num_classes != 2:
            raise ValueError(f"cnn_small: the HIP classifier/loss kernels implement num_classes == 2, got {num_classes}")
        if input_channels != 1:
            raise ValueError(f"cnn_small: input_channels must be 1 (mono spectrogram), got {input_channels}")
        if pretrained:
            raise ValueError("cnn_small has no pretrained weights (pass pretrained=False)")
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        self.stem = nn.Sequential(OrderedDict(conv=nn.Conv2d(1, self.CH, 3, stride=2, padding=1, bias=False),
                                              bn=nn.BatchNorm2d(self.CH)))
        self.blocks = nn.ModuleList([_DSBlock(self.CH) for _ in range(self.N_BLOCKS)])
        if not self.features_only:
            self.classifier = nn.Linear(self.CH, num_classes)
        self.p = float(dropout)
        self.dropout_seed = int(dropout_seed)
        self.dropout_step = 0        # advanced once per training-mode forward (counter of the Philox stream)
        self.sample_offset = 0       # first global sample index of this rank's shard (data parallel)
        # experiment switch (tools/ab input gate, profiles/EXPERIMENTS.md): None = the next batch's input stage starts whenever
        # its stream is free; "fwd" / "bwd" / "mid" = a HIP event recorded before this step's forward / after its loss /
        # between the two halves of its backward, for the input stream to wait on
        self.input_gate, self.gate_event = None, None
        self._pending_tracked = 0
        self._reset_caches()
        self.register_state_dict_pre_hook(CNNSmallWakeword._state_dict_hook)     # (a plain function: the module stays picklable)
        self.register_load_state_dict_pre_hook(CNNSmallWakeword._load_hook)

    # ------------------------------------------------------------------ parameter plumbing
    def _bns(self):
        return [self.stem.bn] + [b for blk in self.blocks for b in (blk.dw_bn, blk.pw_bn)]

    def _ordered(self):
        """Tensors in the C-ABI order of ww_cnn_small_fwd (include/wwhip.h)."""
        out = [self.stem.conv.weight, self.stem.bn.weight, self.stem.bn.bias, self.stem.bn.running_mean,
               self.stem.bn.running_var]
        for blk in self.blocks:
            out += [blk.dw.weight, blk.dw_bn.weight, blk.dw_bn.bias, blk.dw_bn.running_mean, blk.dw_bn.running_var,
                    blk.pw.weight, blk.pw_bn.weight, blk.pw_bn.bias, blk.pw_bn.running_mean, blk.pw_bn.running_var]
        return out + ([None, None] if self.features_only else [self.classifier.weight, self.classifier.bias])

    def _reset_caches(self):
        self._plist = None
        self._pkey = None
        self._pptr = None
        self._gptr = None
        self._flat_grad = None
        self._flat_grad_ext = None
        self._late_offset = 0
        self._flat_param = None
        self._grad_views = None
        self._ws = {}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._reset_caches()
        return r

    def _flush_tracked(self):
        if self._pending_tracked:
            for bn in self._bns():
                bn.num_batches_tracked += self._pending_tracked
            self._pending_tracked = 0

    @staticmethod
    def _state_dict_hook(module, prefix, keep_vars):
        module._flush_tracked()

    @staticmethod
    def _load_hook(module, state_dict, prefix, *args):
        module._pending_tracked = 0             # the loaded num_batches_tracked is the truth: nothing pending carries over

    def _prepare(self, dev):
        tensors = self._ordered()
        key = tuple(0 if t is None else t.data_ptr() for t in tensors)
        if self._pptr is not None and self._pkey == key:
            return
        for t in tensors:
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise nat.NativeError("cnn_small parameters must be contiguous float32 on the input's device")
        # parameters become views of ONE flat fp32 bucket (same order as the gradient bucket): the fused clip+optimizer
        # kernel walks both buckets linearly; state_dict()/load_state_dict() are unaffected (copy_ goes through views)
        plist = [t for t in tensors if isinstance(t, nn.Parameter)]
        n_all = sum(t.numel() for t in plist)
        packed = all(b.data_ptr() == a.data_ptr() + 4 * a.numel() for a, b in zip(plist[:-1], plist[1:]))
        if packed and plist[0].data.untyped_storage().nbytes() >= 4 * (plist[0].storage_offset() + n_all):
            # the parameters already sit back to back in one storage, in this order: an enclosing model (CRNNWakeword,
            # FlatBuckets) has put them into ITS bucket -- adopt that range instead of moving them out again
            flat = torch.empty(0, dtype=torch.float32, device=dev).set_(plist[0].data.untyped_storage(),
                                                                         plist[0].storage_offset(), (n_all,))
        else:
            flat = torch.empty(n_all, dtype=torch.float32, device=dev)
            off = 0
            for t in plist:
                view = flat[off:off + t.numel()].view_as(t)
                view.copy_(t.data)
                t.data = view
                off += t.numel()
        self._flat_param = flat
        key = tuple(0 if t is None else t.data_ptr() for t in tensors)
        self._pkey = key
        self._pptr = nat.ptr_array(tensors)
        self._plist = [t for t in tensors if isinstance(t, nn.Parameter)]
        # no alignment padding: EVERY element of the bucket is rewritten by each backward, so a non-finite step
        # (clip multiplies by NaN) cannot leave poison behind in never-written slots
        sizes = [t.numel() for t in self._plist]
        # one spare float behind the gradients: the data-parallel found_inf flag (written by the loss kernel, summed by the
        # same all-reduce as the gradients, read by the fused optimizer) -- see Trainer._step_native
        self._flat_grad_ext = torch.zeros(sum(sizes) + 1, dtype=torch.float32, device=dev)
        self._flat_grad = self._flat_grad_ext[:-1]
        views, off = {}, 0
        self._late_offset = 0
        for t, n in zip(self._plist, sizes):
            if self.N_BLOCKS >= 3 and t is self.blocks[2].dw.weight:
                self._late_offset = off              # first gradient written by the WW_BWD_LATE half of the backward
            views[id(t)] = self._flat_grad[off:off + t.numel()].view_as(t)
            off += n
        self._grad_views = views
        self._gptr = nat.ptr_array([None if t is None else views.get(id(t)) for t in tensors])

    @property
    def flat_param(self):
        """The flat fp32 parameter bucket every nn.Parameter of this model is a view of (built on first use)."""
        if self._flat_param is None:
            self._prepare(self.classifier.weight.device)
        return self._flat_param

    def grads_in_bucket(self) -> bool:
        """True when every .grad is the model's own view of ``flat_grad`` (the normal case after one backward)."""
        return self._grad_views is not None and all(
            p.grad is not None and p.grad.data_ptr() == self._grad_views[id(p)].data_ptr() for p in self._plist)

    @property
    def flat_grad(self):
        """The one flat fp32 gradient bucket (all-reduce / clip operate on it)."""
        return self._flat_grad

    @property
    def flat_grad_ext(self):
        """``flat_grad`` plus one trailing float, the data-parallel found_inf slot (``flat_grad_ext[-1:]``)."""
        return self._flat_grad_ext

    @property
    def late_offset(self) -> int:
        """``flat_grad[late_offset:]`` = gradients of blocks 2, 3 and the classifier (complete after WW_BWD_LATE)."""
        return self._late_offset

    def set_act_dtype(self, act_dtype):
        """'fp32' (parity mode) or 'bf16' (half the HBM traffic; fp32 arithmetic and statistics)."""
        self.act = nat.act_code(act_dtype)
        self._ws = {}

    def _workspace(self, B, F, T, dev, hold):
        key = (B, F, T, self.act)
        slot = self._ws.get(key)
        if slot is None or slot["busy"]:
            n = nat.cnn_small_workspace_bytes(B, F, T, self.act)
            slot = {"buf": torch.empty(n // 4, dtype=torch.float32, device=dev), "busy": False}
            self._ws[key] = slot
        slot["busy"] = hold
        return slot

    # ------------------------------------------------------------------ launches
    def _check_input(self, x):
        if x.dim() == 3:
            x = x.unsqueeze(1)
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"cnn_small expects (B,1,F,T) features, got {tuple(x.shape)}")
        if not x.is_cuda:
            raise nat.NativeError("cnn_small runs on hand-written HIP kernels only: the input is on "
                                  f"'{x.device}', need an MI355X ('cuda') device -- there is no CPU fallback")
        if x.dtype != torch.float32:
            raise ValueError(f"cnn_small expects float32 features, got {x.dtype}")
        return x.contiguous()

    def _launch_forward(self, x, training):
        dev = x.device
        self._prepare(dev)
        B, _, F, T = x.shape
        slot = self._workspace(B, F, T, dev, hold=training and torch.is_grad_enabled())
        logits = torch.empty((B, 2), dtype=torch.float32, device=dev)
        bn0 = self.stem.bn
        step = self.dropout_step
        nat.cnn_small_fwd(self._pptr, x, slot["buf"], logits, training=training,
                          bn_momentum=bn0.momentum if bn0.momentum is not None else 0.1, bn_eps=bn0.eps,
                          dropout_p=self.p, seed=self.dropout_seed, step=step, sample_offset=self.sample_offset,
                          act=self.act)
        if training:
            self.dropout_step += 1
            self._pending_tracked += 1
        return logits, slot, step

    def _launch_backward(self, x, dlogits, slot, step, mid_hook=None):
        fresh = all(p.grad is None for p in self._plist)
        if fresh and mid_hook is not None:      # two halves; the caller starts reducing the late layers' gradients between them
            for part in (nat.BWD_LATE, nat.BWD_EARLY):
                nat.cnn_small_bwd(self._pptr, self._gptr, x, dlogits.contiguous(), slot["buf"], dropout_p=self.p,
                                  seed=self.dropout_seed, step=step, sample_offset=self.sample_offset, act=self.act, part=part)
                if part == nat.BWD_LATE:
                    mid_hook()
            slot["busy"] = False
            for p in self._plist:
                p.grad = self._grad_views[id(p)]
            return
        if fresh:
            gptr = self._gptr
        else:                                   # accumulate into existing .grad tensors
            tmp = torch.zeros_like(self._flat_grad)
            views, off, tens = {}, 0, self._ordered()
            for t in self._plist:
                views[id(t)] = tmp[off:off + t.numel()].view_as(t)
                off += t.numel()
            gptr = nat.ptr_array([None if t is None else views.get(id(t)) for t in tens])
        nat.cnn_small_bwd(self._pptr, gptr, x, dlogits.contiguous(), slot["buf"], dropout_p=self.p,
                          seed=self.dropout_seed, step=step, sample_offset=self.sample_offset, act=self.act)
        slot["busy"] = False
        for p in self._plist:
            if fresh:
                p.grad = self._grad_views[id(p)]
            elif p.grad is None:
                p.grad = views[id(p)].clone()
            else:
                p.grad.add_(views[id(p)])

    # ------------------------------------------------------------------ CRNN front-end (features_only)
    def _launch_front_forward(self, x, training):
        dev = x.device
        self._prepare(dev)
        B, _, F, T = x.shape
        slot = self._workspace(B, F, T, dev, hold=training and torch.is_grad_enabled())
        seq = torch.empty((B, (T + 1) // 2, self.CH), dtype=torch.float32, device=dev)
        bn0 = self.stem.bn
        nat.cnn_front_fwd(self._pptr, x, slot["buf"], seq, training=training,
                          bn_momentum=bn0.momentum if bn0.momentum is not None else 0.1, bn_eps=bn0.eps, act=self.act)
        if training:
            self._pending_tracked += 1
        return seq, slot

    def _launch_front_backward(self, x, dseq, slot):
        fresh = all(p.grad is None for p in self._plist)
        if fresh:
            gptr, views = self._gptr, None
        else:
            tmp = torch.zeros_like(self._flat_grad)
            views, off, tens = {}, 0, self._ordered()
            for t in self._plist:
                views[id(t)] = tmp[off:off + t.numel()].view_as(t)
                off += t.numel()
            gptr = nat.ptr_array([None if t is None else views.get(id(t)) for t in tens])
        nat.cnn_front_bwd(self._pptr, gptr, x, dseq.contiguous(), slot["buf"], act=self.act)
        slot["busy"] = False
        for p in self._plist:
            if fresh:
                p.grad = self._grad_views[id(p)]
            elif p.grad is None:
                p.grad = views[id(p)].clone()
            else:
                p.grad.add_(views[id(p)])

    def train_step_native(self, x: torch.Tensor, targets: torch.Tensor, criterion, mid_hook=None, found_inf_out=None):
        """forward -> native loss -> backward as three C-ABI calls, without the autograd engine (what Trainer's native step
        uses: the loss kernel already returns dL/dlogits, so nothing needs recording).  Gradients land in the flat bucket
        exactly as after ``criterion(model(x), targets).backward()``; returns the device ``ww_step_stats`` tensor.
        Data parallel: ``mid_hook()`` is called once ``flat_grad[late_offset:]`` is complete (the backward then runs as two
        C-ABI calls); ``found_inf_out`` (float32[1]) also receives the loss kernel's skip flag."""
        x = self._check_input(x)
        if not self.training:
            raise RuntimeError("train_step_native() needs model.train()")
        if any(p.grad is not None for p in self._plist or ()):
            raise RuntimeError("train_step_native() writes fresh gradients: call optimizer.zero_grad(set_to_none=True) first")
        self._prepare(x.device)
        gate = self.input_gate                     # where in this step the NEXT batch's input stage may start (Trainer)
        if gate == "fwd":
            self.gate_event.record()
        logits, slot, step = self._launch_forward(x, training=True)
        stats, dlogits = criterion.native_fwd_bwd(logits, targets, found_inf_out=found_inf_out)
        if gate == "bwd":
            self.gate_event.record()
        if gate == "mid" and mid_hook is None:
            mid_hook = self.gate_event.record
        self._launch_backward(x, dlogits, slot, step, mid_hook=mid_hook)
        return stats

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self._check_input(x)
        if self.features_only:                     # -> (B, ceil(T/2), 64) sequence
            if self.training and torch.is_grad_enabled():
                self._prepare(x.device)
                return _CNNFrontFn.apply(x, self, *self._plist)
            return self._launch_front_forward(x, training=self.training)[0]
        if self.training and torch.is_grad_enabled():
            self._prepare(x.device)
            return _CNNSmallFn.apply(x, self, *self._plist)
        logits, _, _ = self._launch_forward(x, training=self.training)
        return logits


def create_model(architecture: str, num_classes: int = 2, pretrained: bool = False, **kwargs) -> nn.Module:
    """Factory (same signature and error behaviour as the reference's, case-insensitive name)."""
    name = architecture.lower()
    if name == "cnn_small":
        return CNNSmallWakeword(num_classes=num_classes, pretrained=pretrained, dropout=kwargs.get("dropout", 0.3),
                                input_channels=kwargs.get("input_channels", 1),
                                dropout_seed=kwargs.get("dropout_seed", 0), act_dtype=kwargs.get("act_dtype", "fp32"))
    if name == "mobilenetv3":                           # same kwargs as the reference factory (architectures.py:468-474)
        from .mobilenet import MobileNetV3Wakeword
        return MobileNetV3Wakeword(num_classes=num_classes, pretrained=pretrained, dropout=kwargs.get("dropout", 0.3),
                                   input_channels=kwargs.get("input_channels", 1), mode=kwargs.get("mode", "fp32"),
                                   dropout_seed=kwargs.get("dropout_seed", 0))
    if name == "crnn":                                  # not in the reference factory: BASELINE config 5's model
        from .recurrent import CRNNWakeword
        return CRNNWakeword(num_classes=num_classes, hidden_size=kwargs.get("hidden_size", 128),
                            num_layers=kwargs.get("num_layers", 2), bidirectional=kwargs.get("bidirectional", True),
                            dropout=kwargs.get("dropout", 0.3), dropout_seed=kwargs.get("dropout_seed", 0),
                            act_dtype=kwargs.get("act_dtype", "fp32"))
    if name == "gru":                                   # same kwargs as the reference factory (architectures.py:490-498)
        from .recurrent import GRUWakeword
        return GRUWakeword(input_size=kwargs.get("input_size", 40), hidden_size=kwargs.get("hidden_size", 128),
                           num_layers=kwargs.get("num_layers", 2), num_classes=num_classes,
                           bidirectional=kwargs.get("bidirectional", True), dropout=kwargs.get("dropout", 0.3),
                           dropout_seed=kwargs.get("dropout_seed", 0), mode=kwargs.get("mode", "fp32"))
    if name in _REFERENCE_ONLY:
        raise ValueError(f"Architecture '{architecture}' exists in the reference but is outside this build's "
                         f"HIP hot path (DESIGN.md 'Out of scope'). Supported: {_SUPPORTED}")
    raise ValueError(f"Unknown architecture: {architecture}. Supported: {_SUPPORTED}")
