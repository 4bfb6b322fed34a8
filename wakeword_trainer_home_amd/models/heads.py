"""Dense classifier head on the matrix cores.  ``MobileNetV3Head`` has the structure and the
``state_dict`` keys (``0.weight, 0.bias, 3.weight, 3.bias``) of the ``classifier`` the reference installs on
MobileNetV3 (``src/models/architectures.py:105-111``: ``Sequential(Linear(num_features,1024), Hardswish(),
Dropout(dropout), Linear(1024,num_classes))``), so it drops into that slot; the arithmetic is ``ww_linear_mfma_fwd/bwd``
(Hardswish and dropout fused into the first GEMM's epilogue).  The MobileNetV3 body itself is not part of this build yet."""
import torch
import torch.nn as nn

from .. import _native as nat
from .flat_buckets import grad_slot


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mod, step):
        p = mod.dropout if mod.training else 0.0
        need_pre = mod._act != nat.LIN_NONE
        out = nat.linear_mfma_fwd(x, weight, bias, act=mod._act, dropout_p=p, seed=mod.dropout_seed, step=step,
                                  sample_offset=mod.sample_offset, mode=mod.mode, want_pre=need_pre)
        y, pre = out if need_pre else (out, None)
        ctx.save_for_backward(x, weight, pre)
        ctx.mod, ctx.p, ctx.step, ctx.has_bias = mod, p, step, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, pre = ctx.saved_tensors
        mod = ctx.mod
        dx, dw, db = nat.linear_mfma_bwd(x, weight, pre, dy.contiguous(), act=mod._act, dropout_p=ctx.p, seed=mod.dropout_seed,
                                         step=ctx.step, sample_offset=mod.sample_offset, mode=mod.mode,
                                         need_dx=ctx.needs_input_grad[0], need_db=ctx.has_bias,
                                         dw_out=grad_slot(mod.weight), db_out=grad_slot(mod.bias) if ctx.has_bias else None)
        return dx, dw, db, None, None


class MFMALinear(nn.Module):
    """``nn.Linear`` (+ optional Hardswish + Dropout after it) on MFMA.  ``mode``: 'fp32' (fp32 MFMA, parity mode) or
    'bf16' (operands rounded to bf16, fp32 accumulation).  Parameters are ordinary fp32 ``weight`` (out,in) / ``bias``."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, activation: str = None,
                 dropout: float = 0.0, mode="fp32", dropout_seed: int = 0):
        super().__init__()
        if activation not in (None, "hardswish"):
            raise ValueError(f"activation must be None or 'hardswish', got {activation!r}")
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        ref = nn.Linear(in_features, out_features, bias=bias)           # torch's own initialisation
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None
        self.in_features, self.out_features = in_features, out_features
        self._act = nat.LIN_HARDSWISH if activation == "hardswish" else nat.LIN_NONE
        self.dropout, self.dropout_seed = float(dropout), dropout_seed
        self.mode = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}.get(mode, mode)
        nat.act_code(self.mode)
        self.dropout_step = 0          # advanced once per training forward
        self.sample_offset = 0         # global index of the batch's first sample (data parallel shards)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise nat.NativeError("MFMALinear runs on hand-written HIP kernels only: the input is on "
                                  f"'{x.device}', need an MI355X ('cuda') device -- there is no CPU fallback")
        if x.dim() != 2 or x.shape[1] != self.in_features:
            raise ValueError(f"expected input (B,{self.in_features}), got {tuple(x.shape)}")
        step = self.dropout_step
        if self.training and self.dropout > 0:
            self.dropout_step += 1
        return _LinearFn.apply(x.float().contiguous(), self.weight, self.bias, self, step)


class MobileNetV3Head(nn.Sequential):
    """Linear(in_features,1024) -> Hardswish -> Dropout -> Linear(1024,num_classes), indices as in the reference."""

    def __init__(self, in_features: int = 576, hidden: int = 1024, num_classes: int = 2, dropout: float = 0.3,
                 mode="fp32", dropout_seed: int = 0):
        super().__init__(MFMALinear(in_features, hidden, activation="hardswish", dropout=dropout, mode=mode,
                                    dropout_seed=dropout_seed),
                         nn.Identity(),      # Hardswish: fused into module 0's epilogue
                         nn.Identity(),      # Dropout:   fused into module 0's epilogue
                         MFMALinear(hidden, num_classes, mode=mode))
