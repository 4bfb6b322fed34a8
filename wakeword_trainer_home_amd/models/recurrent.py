"""``GRUWakeword`` with the constructor, ``forward`` contract and ``state_dict`` keys of the reference class
(``src/models/architectures.py:198-267``: ``gru.weight_ih_l0`` ... ``gru.bias_hh_l1_reverse``, ``fc.1.weight``, ``fc.1.bias``),
running on ``ww_gru_fwd/bwd`` (one persistent MFMA kernel per layer and direction), ``ww_dropout_bt`` and the MFMA
``fc``.  It also accepts the (B,1,F,T) feature batches the Trainer produces (the reference class takes (B,T,F) only and
cannot be driven by its own Trainer -- SURVEY.md Q3).  Hidden size 128 (the reference default) is the implemented size."""
import math

import torch
import torch.nn as nn

from .. import _native as nat
from .flat_buckets import FlatBuckets, grad_slot
from .heads import MFMALinear


class _GRUStackFn(torch.autograd.Function):
    """All layers and directions of the stack; parameters arrive flat in nn.GRU's ``_flat_weights`` order.  The two
    directions of a layer are independent: the reverse one runs on a second HIP stream beside the forward one (a
    per-direction recurrence occupies B/16 workgroups -- at the per-GPU batches of data-parallel training that is a
    fraction of the 256 CUs), joined before the next layer."""

    @staticmethod
    def _fused(mod, B):
        """One recurrent launch for both directions (bit-identical to the per-direction launches): always while a HIP graph is
        being captured (a fork inside a graph does not run concurrently on this stack), and eagerly up to FUSE_MAX_BATCH rows --
        above it a single direction already fills the 256 CUs (B/16 workgroups, one per CU) and the two-stream form wins by
        running one direction's GEMMs under the other's recurrence (CRNN B=4096: 16.4 vs 16.8 ms)."""
        if mod.num_directions != 2 or not mod.fused_directions:
            return False
        return B <= mod.FUSE_MAX_BATCH or not mod.overlap_directions or torch.cuda.is_current_stream_capturing()

    @staticmethod
    def _par(mod, dev, B):
        return mod.num_directions == 2 and mod.overlap_directions and not _GRUStackFn._fused(mod, B)

    @staticmethod
    def forward(ctx, x, mod, step, *params):
        L, nd, H = mod.num_layers, mod.num_directions, mod.hidden_size
        B, T, _ = x.shape
        dev = x.device
        p = mod.dropout if (mod.training and L > 1) else 0.0
        main = torch.cuda.current_stream(dev)
        side = mod.side_stream(dev) if _GRUStackFn._par(mod, dev, B) else None
        fused = _GRUStackFn._fused(mod, B)
        inputs, workspaces, h_last = [x], [], []
        cur = x
        for k in range(L):
            out = torch.empty((B, T, nd * H), dtype=torch.float32, device=dev)
            ws_k, h_k = [None] * nd, [None] * nd

            def run(d):
                w_ih, w_hh, b_ih, b_hh = params[4 * (k * nd + d):4 * (k * nd + d) + 4]
                ws_k[d] = nat.gru_workspace(B, T, cur.shape[2], H, dev)
                h_k[d] = nat.gru_fwd(cur, w_ih, w_hh, b_ih, b_hh, out[:, :, d * H:(d + 1) * H], ws_k[d], reverse=(d == 1),
                                     mode=mod.mode)
            if fused:                                        # both directions: one recurrent launch (gridDim.y = 2)
                ws_k = [nat.gru_workspace(B, T, cur.shape[2], H, dev) for _ in range(2)]
                h_k = nat.gru_bidir_fwd(cur, [params[4 * (k * nd + d):4 * (k * nd + d) + 4] for d in range(2)], out, ws_k,
                                        mode=mod.mode)
            elif side is not None:
                side.wait_stream(main)                       # `cur` and `out` are ready / allocated
                with torch.cuda.stream(side):
                    run(1)
                run(0)
                main.wait_stream(side)
                ws_k[1].record_stream(main)
                h_k[1].record_stream(main)
            else:
                for d in range(nd):
                    run(d)
            workspaces.append(ws_k)
            if k == L - 1:
                h_last = h_k
            if p > 0 and k + 1 < L:
                out = nat.dropout_bt(out, p, seed=mod.dropout_seed, step=step, sample_offset=mod.sample_offset, stream_id=1 + k)
            cur = out
            if k + 1 < L:
                inputs.append(cur)
        ctx.mod, ctx.step, ctx.p = mod, step, p
        ctx.inputs, ctx.workspaces = inputs, workspaces
        ctx.param_objs = params            # the Parameter objects themselves: their gradient-bucket slots (grad_slot)
        ctx.save_for_backward(*params)
        return torch.cat(h_last, dim=1) if nd == 2 else h_last[0]

    @staticmethod
    def backward(ctx, dh):
        mod, params = ctx.mod, ctx.saved_tensors
        L, nd, H = mod.num_layers, mod.num_directions, mod.hidden_size
        grads = [None] * len(params)
        dh = dh.contiguous()
        dev = dh.device
        main = torch.cuda.current_stream(dev)
        B = ctx.inputs[0].shape[0]
        side = mod.side_stream(dev) if _GRUStackFn._par(mod, dev, B) else None
        fused = _GRUStackFn._fused(mod, B)
        dy = None                                  # gradient of layer k's (dropped-out) output, (B,T,nd*H)
        for k in reversed(range(L)):
            xin = ctx.inputs[k]
            need_dx = k > 0 or ctx.needs_input_grad[0]
            dx = torch.empty_like(xin) if need_dx else None
            dhn = [dh[:, d * H:(d + 1) * H].contiguous() if k == L - 1 else None for d in range(nd)]

            def run(d, dx_d, acc):
                w_ih, w_hh = params[4 * (k * nd + d)], params[4 * (k * nd + d) + 1]
                dyd = dy[:, :, d * H:(d + 1) * H] if dy is not None else None
                g = nat.gru_bwd(xin, w_ih, w_hh, dyd, dhn[d], ctx.workspaces[k][d], reverse=(d == 1), dx=dx_d,
                                accumulate_dx=acc, mode=mod.mode)
                grads[4 * (k * nd + d):4 * (k * nd + d) + 4] = g[:4]
            if fused:
                # gradients born in their slots of the model's flat bucket (autograd adopts them without reading), so the sums
                # of the weight-gradient / bias partials can wait for the ONE flush at the end of the backward pass
                slots = [tuple(grad_slot(q) for q in ctx.param_objs[4 * (k * nd + d):4 * (k * nd + d) + 4]) for d in range(2)]
                have = all(s_ is not None for d_ in slots for s_ in d_)
                g = nat.gru_bidir_bwd(xin, [params[4 * (k * nd + d):4 * (k * nd + d) + 2] for d in range(2)], dy, dhn,
                                      ctx.workspaces[k], dx=dx, mode=mod.mode, outs=slots if have else None,
                                      defer=have and nat.defer_begin(dev))
                for d in range(2):
                    grads[4 * (k * nd + d):4 * (k * nd + d) + 4] = g[d]
            elif side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    dx1 = torch.empty_like(xin) if need_dx else None       # its own dx: the two run concurrently
                    run(1, dx1, False)
                run(0, dx, False)
                main.wait_stream(side)
                for t in grads[4 * (k * nd + 1):4 * (k * nd + 1) + 4]:
                    t.record_stream(main)
                if need_dx:
                    dx1.record_stream(main)
                    dx.add_(dx1)
            else:
                for d in range(nd):
                    run(d, dx, d > 0)
            dy = dx
            if k > 0 and ctx.p > 0:                # backward of the inter-layer dropout: the same mask on the gradient
                dy = nat.dropout_bt(dy, ctx.p, seed=mod.dropout_seed, step=ctx.step, sample_offset=mod.sample_offset,
                                    stream_id=k)
        ctx.workspaces = ctx.inputs = None
        return (dy if ctx.needs_input_grad[0] else None, None, None) + tuple(grads)


class NativeGRU(nn.Module):
    """Parameter container with nn.GRU's names/initialisation; ``forward(x (B,T,I)) -> h_n of the last layer (B, nd*H)``."""
    FUSE_MAX_BATCH = 1024

    def __init__(self, input_size, hidden_size=128, num_layers=2, bidirectional=True, dropout=0.0, dropout_seed=0,
                 mode="fp32"):
        super().__init__()
        self.mode = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}.get(mode, mode)   # matrix type of the projection GEMMs
        nat.act_code(self.mode)
        if hidden_size != 128:
            raise nat.NativeError(f"the HIP GRU kernels implement hidden_size == 128 (the reference default), got {hidden_size}")
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.num_directions = 2 if bidirectional else 1
        self.dropout, self.dropout_seed = float(dropout), dropout_seed
        self.dropout_step, self.sample_offset = 0, 0
        # both directions of a layer in ONE recurrent launch (ww_gru_bidir_*); False: one launch per direction, the reverse one
        # on a second HIP stream when overlap_directions (the round-2 form, kept for A/B measurements)
        self.fused_directions = True
        self.overlap_directions = True
        self._side = {}
        k = 1.0 / math.sqrt(hidden_size)
        self._names = []
        for layer in range(num_layers):
            for d in range(self.num_directions):
                sfx = "_reverse" if d == 1 else ""
                isz = input_size if layer == 0 else hidden_size * self.num_directions
                for name, shape in ((f"weight_ih_l{layer}{sfx}", (3 * hidden_size, isz)),
                                    (f"weight_hh_l{layer}{sfx}", (3 * hidden_size, hidden_size)),
                                    (f"bias_ih_l{layer}{sfx}", (3 * hidden_size,)), (f"bias_hh_l{layer}{sfx}", (3 * hidden_size,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))     # nn.GRU.reset_parameters
                    self._names.append(name)

    def side_stream(self, dev):
        if dev not in self._side:
            self._side[dev] = torch.cuda.Stream(device=dev)
        return self._side[dev]

    def forward(self, x):
        if not x.is_cuda:
            raise nat.NativeError("the GRU runs on hand-written HIP kernels only: the input is on "
                                  f"'{x.device}', need an MI355X ('cuda') device -- there is no CPU fallback")
        if x.dim() != 3 or x.shape[2] != self.input_size:
            raise ValueError(f"expected input (B,T,{self.input_size}), got {tuple(x.shape)}")
        nat.defer_reset(x.device)              # (a previous backward pass that raised midway must not leave its queue behind)
        step = self.dropout_step
        if self.training and self.dropout > 0 and self.num_layers > 1:
            self.dropout_step += 1
        return _GRUStackFn.apply(x.float().contiguous(), self, step, *[getattr(self, n) for n in self._names])


class _HiddenDropout(nn.Module):
    """The ``nn.Dropout`` in front of ``fc`` (architectures.py:239), drawn from the Philox stream (stream 15)."""

    def __init__(self, p, owner):
        super().__init__()
        self.p = float(p)
        self._owner = [owner]          # not a submodule: only read for seed / step / sample offset

    def forward(self, h):
        gru = self._owner[0]
        if not self.training or self.p <= 0:
            return h
        return _DropFn.apply(h, self.p, gru.dropout_seed, gru.fc_step, gru.sample_offset)


class _DropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, p, seed, step, offset):
        ctx.args = (p, seed, step, offset)
        return nat.dropout_bt(h.contiguous()[:, None, :], p, seed=seed, step=step, sample_offset=offset, stream_id=15)[:, 0]

    @staticmethod
    def backward(ctx, g):
        p, seed, step, offset = ctx.args
        dg = nat.dropout_bt(g.contiguous()[:, None, :], p, seed=seed, step=step, sample_offset=offset, stream_id=15)[:, 0]
        return dg, None, None, None, None


class CRNNWakeword(FlatBuckets, nn.Module):
    # FlatBuckets: all 45 parameter tensors are views of one fp32 bucket (the conv front-end adopts its range of it), so the
    # step ends in the fused clip + optimizer launch, data-parallel training all-reduces one tensor, and the whole step can
    # be captured as a HIP graph (Trainer._graph_capture)
    hip_backed = True      # every op is a HIP kernel of this build: the Trainer may run its sync-free step (no host reads)

    """Conv front-end + GRU (BASELINE config 5's model; the reference has no CRNN -- SURVEY.md F4 -- so the topology is this
    build's, made of the reference's own parts): cnn_small's conv stack (stem + 4 depthwise-separable blocks, 64 channels)
    -> BN+ReLU -> mean over the frequency axis -> (B, T/2, 64) -> GRUWakeword's recurrent part (2-layer bidirectional GRU,
    final hidden states concatenated -> Dropout -> Linear).  ``forward((B,1,F,T)) -> (B, num_classes)``."""

    def __init__(self, num_classes: int = 2, hidden_size: int = 128, num_layers: int = 2, bidirectional: bool = True,
                 dropout: float = 0.3, dropout_seed: int = 0, act_dtype: str = "fp32"):
        super().__init__()
        from .architectures import CNNSmallWakeword
        self.front = CNNSmallWakeword(num_classes=2, dropout=0.0, act_dtype=act_dtype, features_only=True)
        self.rnn = _GRUWakewordBase(input_size=CNNSmallWakeword.CH, hidden_size=hidden_size, num_layers=num_layers,
                               num_classes=num_classes, bidirectional=bidirectional, dropout=dropout, dropout_seed=dropout_seed,
                               mode={nat.ACT_BF16: "bf16", nat.ACT_F16: "fp16"}.get(nat.act_code(act_dtype), "fp32"))

    @property
    def sample_offset(self):
        return self.rnn.sample_offset

    @sample_offset.setter
    def sample_offset(self, v):
        self.rnn.sample_offset = v
        self.front.sample_offset = v

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.rnn(self.front(x))


class _GRUWakewordBase(nn.Module):
    hip_backed = True

    def __init__(self, input_size: int = 40, hidden_size: int = 128, num_layers: int = 2, num_classes: int = 2,
                 bidirectional: bool = True, dropout: float = 0.3, dropout_seed: int = 0, mode: str = "fp32"):
        super().__init__()
        self.hidden_size, self.num_layers, self.bidirectional = hidden_size, num_layers, bidirectional
        self.gru = NativeGRU(input_size, hidden_size, num_layers, bidirectional, dropout if num_layers > 1 else 0.0,
                             dropout_seed, mode=mode)
        self.gru.fc_step = 0
        out = hidden_size * 2 if bidirectional else hidden_size
        self.fc = nn.Sequential(_HiddenDropout(dropout, self.gru), MFMALinear(out, num_classes, mode=mode))

    @property
    def sample_offset(self):
        return self.gru.sample_offset

    @sample_offset.setter
    def sample_offset(self, v):
        self.gru.sample_offset = v

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 4:                                    # (B,1,F,T) feature batch -> (B,T,F)
            if x.shape[1] != 1:
                raise ValueError(f"expected (B,1,F,T) features or (B,T,F) sequences, got {tuple(x.shape)}")
            x = x[:, 0].transpose(1, 2)
        self.gru.fc_step = self.gru.dropout_step            # one Philox step per training forward, shared by all masks
        h = self.gru(x.contiguous())
        if self.training and self.gru.dropout == 0 and self.fc[0].p > 0:
            self.gru.dropout_step += 1                      # single-layer stacks: the fc dropout alone advances the stream
        return self.fc(h)


class GRUWakeword(FlatBuckets, _GRUWakewordBase):
    """The reference's class name; as a top-level model its parameters live in one flat bucket (fused clip + optimizer, one
    all-reduce, graph-capturable step).  Inside ``CRNNWakeword`` the plain base is used -- the CRNN owns the bucket."""
