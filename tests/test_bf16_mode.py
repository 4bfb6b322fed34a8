"""GPU: bf16 activation-storage mode (BASELINE config 2).  Arithmetic and statistics stay fp32; only what is
written to / read from HBM is bf16 (round-to-nearest-even).  Tolerances are bf16-sized and stated per test;
the 1e-3 loss-parity claim belongs to the fp32 mode (tests/test_trainer_gpu.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import load_trace, make_inputs
from tests.test_hip_kernels import cu, nhwc, rel_err, DEV, _bn_tensors

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPS_BF16 = 2.0 ** -8          # one RNE rounding is <= 2^-9 relative; 2^-8 leaves room for fp32 order effects


@pytest.fixture(scope="module")
def nat():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wakeword_trainer_home_amd import _native
    _native.load()
    return _native


def rb(t):
    """round a float64/32 tensor to bf16 and back (what the device stores)."""
    return t.float().to(BF).double()


@pytest.mark.parametrize("kind", ["dw", "pw"])
@pytest.mark.parametrize("B,H,W", [(3, 20, 76), (2, 7, 25), (5, 17, 10)])
def test_conv_fwd_layers_bf16(nat, kind, B, H, W):
    g = torch.Generator().manual_seed(31)
    y_in = rb(torch.randn(B, 64, H, W, generator=g, dtype=torch.float64))
    s_in = torch.rand(64, generator=g, dtype=torch.float64) + 0.5
    t_in = torch.randn(64, generator=g, dtype=torch.float64) * 0.5
    a = torch.relu(y_in * s_in[None, :, None, None] + t_in[None, :, None, None])
    if kind == "dw":
        w = torch.randn(64, 1, 3, 3, generator=g, dtype=torch.float64) * 0.3
        ref = F.conv2d(a, w, padding=1, groups=64)
    else:
        w = torch.randn(64, 64, 1, 1, generator=g, dtype=torch.float64) * 0.2
        ref = F.conv2d(a, w)
    gamma, beta, rm, rv = _bn_tensors(2)
    ga, be, rm_g, rv_g = cu(gamma), cu(beta), cu(rm), cu(rv)
    bn = nat.make_bn(ga, be, rm_g, rv_g)
    fn = nat.dwconv3x3_fwd if kind == "dw" else nat.pwconv1x1_fwd
    y, ss, mr = fn(cu(nhwc(y_in), BF), cu(torch.cat([s_in, t_in])), cu(w), bn, nat.layer_scratch(DEV))
    assert y.dtype == BF
    yd = y.float().cpu().double()
    # depthwise: one rounding (the stored output).  pointwise (bf16 MFMA): activations and weights are rounded to
    # bf16 as MFMA operands as well -> three roundings
    tol = EPS_BF16 * (3 if kind == "pw" else 1)
    assert (yd - nhwc(ref)).abs().max() <= tol * nhwc(ref).abs().max()
    # the statistics describe exactly the stored (rounded) tensor
    mean = yd.mean(dim=(0, 1, 2))
    var = yd.var(dim=(0, 1, 2), unbiased=False)
    assert np.abs(mr[:64].cpu().numpy() - mean.numpy()).max() < 1e-5 * (mean.abs().max().item() + 1)
    assert rel_err(mr[64:].cpu(), 1.0 / torch.sqrt(var + 1e-5)) < 2e-5


@pytest.mark.parametrize("kind", ["dw", "pw"])
@pytest.mark.parametrize("B,H,W", [(3, 20, 76), (2, 7, 25)])
def test_conv_bwd_layers_bf16(nat, kind, B, H, W):
    from tests.test_hip_kernels import _coef_from
    gen = torch.Generator().manual_seed(41)
    y_in = rb(torch.randn(B, 64, H, W, generator=gen, dtype=torch.float64))
    bn_in, bn_out = torch.nn.BatchNorm2d(64).double(), torch.nn.BatchNorm2d(64).double()
    with torch.no_grad():
        for bn in (bn_in, bn_out):
            bn.weight.copy_(torch.rand(64, generator=gen, dtype=torch.float64) + 0.5)
            bn.bias.copy_(torch.randn(64, generator=gen, dtype=torch.float64) * 0.3)
    shape = (64, 1, 3, 3) if kind == "dw" else (64, 64, 1, 1)
    w = (torch.randn(*shape, generator=gen, dtype=torch.float64) * 0.25).requires_grad_(True)
    z_in = bn_in(y_in)
    z_in.retain_grad()
    a = torch.relu(z_in)
    y_raw = F.conv2d(a, w, padding=1, groups=64) if kind == "dw" else F.conv2d(a, w)
    y = y_raw + (rb(y_raw.detach()) - y_raw.detach())           # forward stores y rounded; gradient passes through
    z = bn_out(y)
    g = rb(torch.randn(B, 64, H, W, generator=gen, dtype=torch.float64) * (torch.rand(B, 64, H, W, generator=gen) > 0.4))
    (z * g).sum().backward()
    coef, _, _ = _coef_from(g, y.detach(), bn_out.weight.detach())
    mean_in = y_in.mean(dim=(0, 2, 3))
    rstd_in = 1.0 / torch.sqrt(y_in.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    scale_in = bn_in.weight.detach() * rstd_in
    ss_in = torch.cat([scale_in, bn_in.bias.detach() - mean_in * scale_in])
    args = dict(y_out=cu(nhwc(y.detach()), BF), coef=cu(coef), y_in=cu(nhwc(y_in), BF), ss_in=cu(ss_in),
                mr_in=cu(torch.cat([mean_in, rstd_in])), gamma_in=cu(bn_in.weight.detach()), w=cu(w.detach()),
                scratch=nat.layer_scratch(DEV))
    if kind == "dw":
        g_in, dw, coef_in, dgamma, dbeta = nat.dwconv3x3_bwd(cu(nhwc(g), BF), **args)
    else:
        g_in, dw, coef_in, dgamma, dbeta = nat.pwconv1x1_bwd(cu(nhwc(g), BF), None, ss_out=None, **args)
    assert g_in.dtype == BF
    gi = g_in.float().cpu().double()
    tol = EPS_BF16 * (3 if kind == "pw" else 1)       # pointwise: dy and W are bf16 MFMA operands
    assert (gi - nhwc(z_in.grad)).abs().max() <= tol * z_in.grad.abs().max()
    # pointwise: dy and relu(bn(y_in)) are rounded to bf16 for the MFMA (fp32 accumulation); depthwise: fp32 operands
    assert rel_err(dw.cpu().reshape(-1), w.grad.reshape(-1)) < (4e-3 if kind == "pw" else 1e-4)
    # sums are taken over the ROUNDED g_in: compare with sums of the device tensor itself
    yhat_in = (nhwc(y_in) - mean_in) * rstd_in
    assert rel_err(dbeta.cpu(), gi.sum(dim=(0, 1, 2))) < 1e-4
    assert rel_err(dgamma.cpu(), (gi * yhat_in).sum(dim=(0, 1, 2))) < 1e-4


@pytest.mark.parametrize("B,Fd,T,p,min_cos", [(4, 40, 151, 0.3, 0.995), (6, 13, 50, 0.0, 0.995),
                                              (5, 8, 12, 0.0, 0.98)])   # last: 4 x 6 maps (HW < one pixel tile), 120 pixels per channel
def test_cnn_small_bf16_close_to_fp32_oracle(nat, B, Fd, T, p, min_cos):
    """whole model in bf16 storage vs the float64 oracle: logits within 3e-2 of their scale, gradient direction
    cos > 0.995 (9 layers x 2 roundings per value of 2^-9 relative each)."""
    from oracle.cnn_small import CNNSmallOracle
    from wakeword_trainer_home_amd.models import create_model
    torch.manual_seed(5)
    oracle = CNNSmallOracle(dropout=p, dropout_seed=3).double()
    model = create_model("cnn_small", dropout=p, dropout_seed=3, act_dtype="bf16")
    model.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()})
    model.to(DEV).train()
    oracle.train()
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(B, 1, Fd, T, generator=gen, dtype=torch.float64) * 2 - 4
    dlog = torch.randn(B, 2, generator=gen, dtype=torch.float64) / B
    out = model(x.float().to(DEV))
    out.backward(dlog.float().to(DEV))
    ref = oracle(x)
    ref.backward(dlog)
    assert (out.detach().cpu().double() - ref.detach()).abs().max() < 3e-2 * max(ref.abs().max().item(), 1.0)
    gn = torch.cat([q.grad.flatten().cpu().double() for q in model.parameters()])
    go = torch.cat([q.grad.flatten() for q in oracle.parameters()])
    cos = (gn @ go / (gn.norm() * go.norm())).item()
    assert cos > min_cos, cos
    # the last pointwise layer's weight gradient comes straight out of the pooled-gradient variant of the backward kernel
    # (per-pixel image lookup when an image is smaller than a tile): 0.993 at every size measured
    ga, gb = model.blocks[3].pw.weight.grad.flatten().cpu().double(), oracle.blocks[3].pw.weight.grad.flatten()
    assert (ga @ gb / (ga.norm() * gb.norm())).item() > 0.99
    model.eval()
    oracle.eval()
    with torch.no_grad():
        e = model(x.float().to(DEV)).cpu().double()
    assert (e - oracle(x)).abs().max() < 3e-2 * max(oracle(x).abs().max().item(), 1.0)


def test_trainer_bf16_tracks_reference_trace(golden_dir, tmp_path):
    """mixed_precision=True selects bf16 storage; per-step loss stays within 5e-4 of the REAL reference Trainer's fp32 trace
    (src/training/trainer.py:165-203 run on CPU, tests/golden/g2_*; measured on MI355X: 4.5e-5 -- the bound is 10x that, inside
    the 1e-3 the fp32 mode is held to)."""
    from wakeword_trainer_home_amd.config import WakewordConfig
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    meta, tr = load_trace(golden_dir, "default_b128")
    cfg = WakewordConfig()
    for sec in ("loss", "optimizer", "training"):
        for k, v in meta["cfg"][sec].items():
            if hasattr(getattr(cfg, sec), k):
                setattr(getattr(cfg, sec), k, v)
    cfg.model.architecture, cfg.optimizer.mixed_precision = "cnn_small", True
    model = create_model("cnn_small", dropout=0.0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tr["init"].items()})
    xtr, ytr = make_inputs(meta["train_seed"], meta["n_train"])
    xva, yva = make_inputs(meta["val_seed"], meta["n_val"])
    DL, TD = torch.utils.data.DataLoader, torch.utils.data.TensorDataset
    t = Trainer(model, DL(TD(xtr, ytr), batch_size=128), DL(TD(xva, yva), batch_size=128), cfg, checkpoint_dir=tmp_path,
                device=DEV)
    from wakeword_trainer_home_amd import _native
    assert model.act == _native.ACT_BF16
    losses = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: losses.append(l)})())
    t.train()
    d = np.abs(np.array(losses) - tr["step_loss"])
    assert d.max() < 5e-4, d
    print(f"bf16 storage: max |loss - ref| = {d.max():.2e}")
