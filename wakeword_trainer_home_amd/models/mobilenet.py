"""``MobileNetV3Wakeword`` (``src/models/architectures.py:68-123``: torchvision's ``mobilenet_v3_small`` with a one-channel
stem conv and the ``Linear(576,1024) -> Hardswish -> Dropout -> Linear(1024, num_classes)`` classifier) on the native
channels-last layer library (``ww_bn_act_*``, ``ww_dwconv_nhwc_*``, squeeze-excitation pieces) and the matrix cores
(``ww_linear_mfma_*`` for every 1x1 convolution, SE FC and the classifier).  The module tree reproduces torchvision's, so the
``state_dict`` keys are the reference model's (``mobilenet.features.4.block.2.fc1.weight`` ...).  torchvision itself is not
available in this build environment: the architecture is restated from its published definition (``oracle/mobilenetv3.py``
carries the same restatement in plain torch.nn) -- parity with torchvision is unpinned, parity with torch.nn is tested.
First version: unfused fp32 layers (correctness and coverage before speed)."""
import torch
import torch.nn as nn

from .. import _native as nat
from .flat_buckets import FlatBuckets, grad_slot
from .heads import MobileNetV3Head

SMALL_CONF = ((16, 3, 16, 16, True, "RE", 2), (16, 3, 72, 24, False, "RE", 2), (24, 3, 88, 24, False, "RE", 1),
              (24, 5, 96, 40, True, "HS", 2), (40, 5, 240, 40, True, "HS", 1), (40, 5, 240, 40, True, "HS", 1),
              (40, 5, 120, 48, True, "HS", 1), (48, 5, 144, 48, True, "HS", 1), (48, 5, 288, 96, True, "HS", 2),
              (96, 5, 576, 96, True, "HS", 1), (96, 5, 576, 96, True, "HS", 1))
LAST_CONV, LAST_CHANNEL = 576, 1024
_ACT = {None: nat.LIN_NONE, "RE": nat.LIN_RELU, "HS": nat.LIN_HARDSWISH}


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


# ------------------------------------------------------------------------------------------ autograd wrappers
class _PWFn(torch.autograd.Function):          # 1x1 convolution on (B,H,W,Cin) = one GEMM over the pixels
    @staticmethod
    def forward(ctx, x, w4, bias, act, mode):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        w2 = w4.reshape(w4.shape[0], -1)
        need_pre = act != nat.LIN_NONE
        out = nat.linear_mfma_fwd(x2, w2, bias, act=act, mode=mode, want_pre=need_pre)
        y, pre = out if need_pre else (out, None)
        ctx.save_for_backward(x2, w2, pre)
        ctx.act, ctx.mode, ctx.shp, ctx.wshape, ctx.has_bias = act, mode, shp, w4.shape, bias is not None
        return y.reshape(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w2, pre = ctx.saved_tensors
        dx, dw, db = nat.linear_mfma_bwd(x2, w2, pre, dy.reshape(-1, w2.shape[0]).contiguous(), act=ctx.act, mode=ctx.mode,
                                         need_dx=ctx.needs_input_grad[0], need_db=ctx.has_bias)
        return (dx.reshape(ctx.shp) if dx is not None else None), dw.reshape(ctx.wshape), db, None, None


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mod, act):
        Cn = x.shape[-1]
        bn = nat.make_bn(gamma, beta, mod.running_mean, mod.running_var, momentum=mod.momentum, eps=mod.eps,
                         training=mod.training)
        y, ss, mr = nat.bn_act_fwd(x.contiguous(), bn, act, Cn)
        if mod.training:
            mod._pending_tracked += 1          # host-side count, folded into the buffer when the state is read
        ctx.save_for_backward(x, ss, mr)
        ctx.act, ctx.training, ctx.Cn = act, mod.training, Cn
        return y

    @staticmethod
    def backward(ctx, da):
        x, ss, mr = ctx.saved_tensors
        dx, dg, db = nat.bn_act_bwd(x.contiguous(), da.contiguous(), ss, mr, ctx.act, ctx.training, ctx.Cn)
        return dx, dg, db, None, None


class _DWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, k, stride):
        ctx.save_for_backward(x, w)
        ctx.k, ctx.stride = k, stride
        return nat.dwconv_nhwc_fwd(x.contiguous(), w.contiguous(), k, stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw = nat.dwconv_nhwc_bwd(x.contiguous(), w.contiguous(), dy.contiguous(), ctx.k, ctx.stride)
        return dx, dw, None, None


class _PoolFn(torch.autograd.Function):       # (B,H,W,C) -> (B,C) mean
    @staticmethod
    def forward(ctx, x):
        B, H, W, Cn = x.shape
        ctx.shp = (B, H * W, Cn)
        ctx.full = x.shape
        return nat.pool_hw_fwd(x.reshape(B, H * W, Cn))

    @staticmethod
    def backward(ctx, ds):
        return nat.scale_pool_bwd(None, None, ds.contiguous(), ctx.shp).reshape(ctx.full)


class _ScaleFn(torch.autograd.Function):      # y[b,h,w,c] = x[b,h,w,c] * gate[b,c]
    @staticmethod
    def forward(ctx, x, gate):
        B, H, W, Cn = x.shape
        ctx.save_for_backward(x, gate)
        return nat.scale_bc_fwd(x.reshape(B, H * W, Cn), gate.contiguous()).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x, gate = ctx.saved_tensors
        B, H, W, Cn = x.shape
        dy3, x3 = dy.contiguous().reshape(B, H * W, Cn), x.reshape(B, H * W, Cn)
        dx = nat.scale_pool_bwd(dy3, gate.contiguous(), None, (B, H * W, Cn)).reshape(x.shape)
        return dx, nat.scale_bc_bwd_gate(x3, dy3)


class _StemFn(torch.autograd.Function):       # Conv2d(1, C, 3, stride 2, pad 1): patches + GEMM; (B,1,H,W) -> (B,Ho,Wo,C)
    @staticmethod
    def forward(ctx, x, w4, mode):
        B, _, H, W = x.shape
        cols = nat.im2col3x3s2(x.reshape(B, H, W).contiguous())
        w2 = w4.reshape(w4.shape[0], 9)
        ctx.save_for_backward(cols, w2)
        ctx.mode, ctx.wshape = mode, w4.shape
        return nat.linear_mfma_fwd(cols, w2, None, mode=mode).reshape(B, (H + 1) // 2, (W + 1) // 2, w4.shape[0])

    @staticmethod
    def backward(ctx, dy):
        cols, w2 = ctx.saved_tensors
        _, dw, _ = nat.linear_mfma_bwd(cols, w2, None, dy.reshape(-1, w2.shape[0]).contiguous(), mode=ctx.mode, need_dx=False,
                                       need_db=False)
        return None, dw.reshape(ctx.wshape), None


class _ConvBNActFn(torch.autograd.Function):
    """Conv2dNormActivation as ONE autograd node (conv -> BatchNorm -> activation): half the Python/autograd overhead of the
    three-node form, which is what bounds the step at the per-GPU batch of data-parallel training.  kind: stem | dw | pw."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, conv, bn, kind, act, mode, residual=None):
        bnp = nat.make_bn(gamma, beta, bn.running_mean, bn.running_var, momentum=bn.momentum, eps=bn.eps, training=bn.training)
        direct_stem = kind == "stem" and bn.training and nat.stem_supported(w.shape[0])
        if direct_stem:
            # the one-channel stem as a direct convolution (no patch tensor; its kernel leaves the BatchNorm statistics)
            B, _, H, W = x.shape
            src = x.reshape(B, H, W).contiguous()
            kind = "stem_direct"
        elif kind == "stem":
            B, _, H, W = x.shape
            src = nat.im2col3x3s2(x.reshape(B, H, W).contiguous())
            w2, oshape = w.reshape(w.shape[0], 9), (B, (H + 1) // 2, (W + 1) // 2, w.shape[0])
        elif kind == "dw":
            src = x.contiguous()
        else:
            src = x.reshape(-1, x.shape[-1])
            w2, oshape = w.reshape(w.shape[0], -1), (*x.shape[:-1], w.shape[0])
        if bn.training:
            # training: the convolution's own kernel leaves the BatchNorm statistics partials and the apply pass finishes them
            if kind == "dw":
                y, a, ss, mr = nat.dwconv_bn_act_fwd(src, w.contiguous(), conv.k, conv.stride, bnp, act)
            elif kind == "stem_direct":
                y, a, ss, mr = nat.stem3x3s2_bn_act_fwd(src, w.contiguous(), bnp, act)
            else:
                # (an inverted-residual block's skip connection is added in the same apply pass)
                res2 = residual.reshape(-1, w2.shape[0]).contiguous() if residual is not None else None
                y, a, ss, mr = nat.conv1x1_bn_act_fwd(src, w2, bnp, act, mode, residual=res2)
                y, a = y.reshape(oshape), a.reshape(oshape)
                residual = None
            bn._pending_tracked += 1           # host-side count, folded into the buffer when the state is read
        else:
            if kind == "dw":
                y = nat.dwconv_nhwc_fwd(src, w.contiguous(), conv.k, conv.stride)
            else:
                y = nat.linear_mfma_fwd(src, w2, None, mode=mode).reshape(oshape)
            a, ss, mr = nat.bn_act_fwd(y, bnp, act, y.shape[-1])
        if residual is not None:               # eval mode / depthwise: the add is its own pass
            a = nat.add_f32(a, residual.contiguous())
        Cn = y.shape[-1]
        ctx.save_for_backward(src, w, y, ss, mr)
        ctx.params = (w, gamma, beta)          # the Parameter objects themselves: their gradient-bucket slots (grad_slot)
        ctx.meta = (kind, act, mode, bn.training, conv.k, conv.stride, x.shape, Cn)
        return a

    @staticmethod
    def backward(ctx, da):
        src, w, y, ss, mr = ctx.saved_tensors
        kind, act, mode, training, k, stride, xshape, Cn = ctx.meta
        wp, gamma, beta = ctx.params
        # parameter gradients are written straight into their slots of the model's flat bucket where there is one (grad_slot)
        dy, dgamma, dbeta = nat.bn_act_bwd(y, da.contiguous(), ss, mr, act, training, Cn, dgamma_out=grad_slot(gamma),
                                           dbeta_out=grad_slot(beta))
        # the weight gradient is born in its bucket slot, which autograd adopts without reading: the "sum the partials" launch
        # of its kernel can wait for the ONE flush at the end of the backward pass (nat.defer_begin)
        slot = grad_slot(wp)
        defer = slot is not None and nat.defer_begin(y.device)
        if kind == "dw":
            dx, dw = nat.dwconv_nhwc_bwd(src, w.contiguous(), dy, k, stride, need_dx=ctx.needs_input_grad[0], dw_out=slot, defer=defer)
        elif kind == "stem_direct":
            dx, dw = None, nat.stem3x3s2_bwd_dw(src, dy, w.shape, dw_out=slot, defer=defer)
        else:
            need_dx = kind == "pw" and ctx.needs_input_grad[0]
            w2 = w.reshape(w.shape[0], -1)
            dx, dw, _ = nat.linear_mfma_bwd(src, w2, None, dy.reshape(-1, Cn), mode=mode, need_dx=need_dx, need_db=False,
                                            dw_out=slot, defer=defer)
            dx = dx.reshape(xshape) if dx is not None else None
            dw = dw.reshape(w.shape)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, (da if ctx.needs_input_grad[9] else None)


class _SEFn(torch.autograd.Function):
    """Squeeze-excitation as one node and THREE launches (``ww_se_fwd`` / ``ww_se_bwd``: a workgroup owns whole images, so the
    pool, both FCs, the gate and the scaling are one kernel, and dx = dy*gate + dpool/HW another; fp32 FMA in every matrix
    mode -- the FCs are B x C x C/4).  Shapes outside the kernels' limits compose the block from the layer library's pieces
    (pool -> matrix-core FC+ReLU -> FC+Hardsigmoid -> scale)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mode):
        B, H, W, Cn = x.shape
        x3 = x.reshape(B, H * W, Cn)
        w1m, w2m = w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1)
        ctx.mode, ctx.xshape, ctx.params = mode, x.shape, (w1, b1, w2, b2)
        ctx.fused = nat.se_supported(Cn, w1m.shape[0], x3, w1m, w2m)
        if ctx.fused:
            y, s, pre1, pre2 = nat.se_fwd(x3.contiguous(), w1m, b1, w2m, b2)
            ctx.save_for_backward(x3, s, pre1, pre2, w1, w2)
            return y.reshape(x.shape)
        s = nat.pool_hw_fwd(x3)
        h, pre1 = nat.linear_mfma_fwd(s, w1m, b1, act=nat.LIN_RELU, mode=mode, want_pre=True)
        g, pre2 = nat.linear_mfma_fwd(h, w2m, b2, act=nat.LIN_HARDSIGMOID, mode=mode, want_pre=True)
        ctx.save_for_backward(x3, s, h, pre1, pre2, g, w1, w2)
        return nat.scale_bc_fwd(x3, g).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        if ctx.fused:
            x3, s, pre1, pre2, w1, w2 = ctx.saved_tensors
            dy3 = dy.contiguous().reshape(x3.shape)
            w1p, b1, w2p, b2 = ctx.params
            dx, dw1, db1, dw2, db2 = nat.se_bwd(x3.contiguous(), dy3, s, pre1, pre2, w1.reshape(w1.shape[0], -1),
                                                w2.reshape(w2.shape[0], -1),
                                                outs=(grad_slot(w1p), grad_slot(b1), grad_slot(w2p), grad_slot(b2)))
            return dx.reshape(ctx.xshape), dw1.reshape(w1.shape), db1, dw2.reshape(w2.shape), db2, None
        x3, s, h, pre1, pre2, g, w1, w2 = ctx.saved_tensors
        B, HW, Cn = x3.shape
        dy3 = dy.contiguous().reshape(B, HW, Cn)
        dg = nat.scale_bc_bwd_gate(x3, dy3)
        dh, dw2, db2 = nat.linear_mfma_bwd(h, w2.reshape(w2.shape[0], -1), pre2, dg, act=nat.LIN_HARDSIGMOID, mode=ctx.mode)
        ds, dw1, db1 = nat.linear_mfma_bwd(s, w1.reshape(w1.shape[0], -1), pre1, dh, act=nat.LIN_RELU, mode=ctx.mode)
        dx = nat.scale_pool_bwd(dy3, g, ds, (B, HW, Cn)).reshape(ctx.xshape)
        return dx, dw1.reshape(w1.shape), db1, dw2.reshape(w2.shape), db2, None


# ------------------------------------------------------------------------------------------ modules (torchvision's tree)
class _Conv(nn.Module):
    """Parameter holder shaped like nn.Conv2d(bias=False): ``weight`` (Cout, Cin/groups, k, k)."""

    def __init__(self, cin, cout, k, stride=1, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out")          # torchvision's MobileNetV3 initialisation
        self.k, self.stride, self.groups = k, stride, groups


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = 0.001, 0.01
        # one training forward = one increment of num_batches_tracked; as a device op that was a kernel launch per BatchNorm
        # layer per step (34 of the step's ~540 launches, 2.5 % of its GPU time) for a number nothing on the device reads
        self._pending_tracked = 0
        self.register_state_dict_pre_hook(_BN._state_dict_hook)      # (plain functions: the module stays picklable)
        self.register_load_state_dict_pre_hook(_BN._load_hook)

    def _flush_tracked(self):
        if self._pending_tracked:
            self.num_batches_tracked += self._pending_tracked
            self._pending_tracked = 0

    @staticmethod
    def _state_dict_hook(module, prefix, keep_vars):
        module._flush_tracked()

    @staticmethod
    def _load_hook(module, state_dict, prefix, *args):
        module._pending_tracked = 0             # the loaded num_batches_tracked is the truth: nothing pending carries over


class ConvBNAct(nn.Sequential):
    """Conv2dNormActivation: children ``0`` (conv), ``1`` (BatchNorm2d); the activation is fused into the BN pass."""

    def __init__(self, cin, cout, k, stride=1, groups=1, act=None, mode=torch.float32, stem=False):
        super().__init__(_Conv(cin, cout, k, stride, groups), _BN(cout))
        self.act, self.mode, self.stem, self.depthwise = _ACT[act], mode, stem, groups > 1

    def forward(self, x, residual=None):
        conv, bn = self[0], self[1]
        kind = "stem" if self.stem else ("dw" if self.depthwise else "pw")
        return _ConvBNActFn.apply(x, conv.weight, bn.weight, bn.bias, conv, bn, kind, self.act, self.mode, residual)


class _FC(nn.Module):
    """nn.Conv2d(cin, cout, 1) parameter holder (the SE block's fc1 / fc2)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))
        nn.init.kaiming_normal_(self.weight, mode="fan_out")
        self.bias = nn.Parameter(torch.zeros(cout))


class SqueezeExcitation(nn.Module):
    def __init__(self, c, cs, mode):
        super().__init__()
        self.fc1, self.fc2, self.mode = _FC(c, cs), _FC(cs, c), mode

    def forward(self, x):
        return _SEFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.mode)


class InvertedResidual(nn.Module):
    def __init__(self, cin, k, exp, cout, use_se, act, stride, mode):
        super().__init__()
        layers = []
        if exp != cin:
            layers.append(ConvBNAct(cin, exp, 1, act=act, mode=mode))
        layers.append(ConvBNAct(exp, exp, k, stride, groups=exp, act=act, mode=mode))
        if use_se:
            layers.append(SqueezeExcitation(exp, _make_divisible(exp // 4, 8), mode))
        layers.append(ConvBNAct(exp, cout, 1, act=None, mode=mode))
        self.block = nn.Sequential(*layers)
        self.use_res_connect = stride == 1 and cin == cout

    def forward(self, x):
        if not self.use_res_connect:
            return self.block(x)
        h = x
        for layer in list(self.block)[:-1]:
            h = layer(h)
        return self.block[-1](h, residual=x)       # the projection layer's BatchNorm pass adds the skip connection


class _MobileNet(nn.Module):
    def __init__(self, num_classes, dropout, mode, dropout_seed):
        super().__init__()
        feats = [ConvBNAct(1, 16, 3, 2, act="HS", mode=mode, stem=True)]
        feats += [InvertedResidual(*c, mode=mode) for c in SMALL_CONF]
        feats.append(ConvBNAct(SMALL_CONF[-1][3], LAST_CONV, 1, act="HS", mode=mode))
        self.features = nn.Sequential(*feats)
        self.classifier = MobileNetV3Head(LAST_CONV, LAST_CHANNEL, num_classes, dropout=dropout, mode=mode,
                                          dropout_seed=dropout_seed)


class MobileNetV3Wakeword(FlatBuckets, nn.Module):
    hip_backed = True
    prefers_hip_graph = True      # ~340 short launches per step: host-bound when issued eagerly (Trainer: training.hip_graph_auto)

    def __init__(self, num_classes: int = 2, pretrained: bool = False, dropout: float = 0.3, input_channels: int = 1,
                 mode="fp32", dropout_seed: int = 0):
        super().__init__()
        if pretrained:
            raise ValueError("pretrained ImageNet weights cannot be downloaded in this build (pass pretrained=False)")
        if input_channels != 1:
            raise ValueError(f"mobilenetv3: the native stem takes one-channel spectrograms, got input_channels={input_channels}")
        m = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}.get(mode, mode)
        nat.act_code(m)
        self.mobilenet = _MobileNet(num_classes, dropout, m, dropout_seed)

    @property
    def sample_offset(self):
        return self.mobilenet.classifier[0].sample_offset

    @sample_offset.setter
    def sample_offset(self, v):
        self.mobilenet.classifier[0].sample_offset = v

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise nat.NativeError("mobilenetv3 runs on hand-written HIP kernels only: the input is on "
                                  f"'{x.device}', need an MI355X ('cuda') device -- there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"mobilenetv3 expects (B,1,F,T) features, got {tuple(x.shape)}")
        nat.defer_reset(x.device)              # (a previous backward pass that raised midway must not leave its queue behind)
        h = self.mobilenet.features(x.float().contiguous())          # (B,H,W,C) channels-last all the way
        return self.mobilenet.classifier(_PoolFn.apply(h))
