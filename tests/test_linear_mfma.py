"""MFMA dense layers / the MobileNetV3 classifier head against torch's own Linear / Hardswish on the
CPU (float64), with the build's Philox dropout mask applied explicitly (oracle/mlp_head.py).
fp32 mode: relative error <= 2e-6 of the tensor's scale (fp32 MFMA accumulation order vs float64).
bf16 mode: compared with the oracle's restatement of that mode (operands rounded to bf16, exact products, wide sums):
<= 2e-5; and against the float64 result within 3 bf16 roundings (2**-8 each) of the accumulated magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.mark.parametrize("M,K,N", [(2048, 576, 1024), (77, 576, 1024), (130, 40, 2), (1, 7, 3), (64, 64, 64), (200, 1024, 2)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_linear_fwd_bwd_matches_torch(M, K, N, mode):
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    md = torch.float32 if mode == "fp32" else torch.bfloat16
    y = nat.linear_mfma_fwd(x.to(DEV), w.to(DEV), b.to(DEV), mode=md)
    dx, dw, db = nat.linear_mfma_bwd(x.to(DEV), w.to(DEV), None, dy.to(DEV), mode=md)
    r = (lambda t: t.bfloat16().double()) if mode == "bf16" else (lambda t: t.double())
    y_ref = r(x) @ r(w).t() + b.double()
    dx_ref = r(dy) @ r(w)
    dw_ref = r(dy).t() @ r(x)
    db_ref = dy.double().sum(0)
    tol = 2e-6 if mode == "fp32" else 2e-5
    assert _rel(y.cpu().double(), y_ref) <= tol
    assert _rel(dx.cpu().double(), dx_ref) <= tol
    assert _rel(dw.cpu().double(), dw_ref) <= tol
    assert _rel(db.cpu().double(), db_ref) <= 2e-6
    if mode == "bf16":      # and the mode itself stays within three bf16 roundings of the exact product
        y64 = x.double() @ w.double().t() + b.double()
        assert _rel(y.cpu().double(), y64) <= 3 * 2.0 ** -8


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("B", [2048, 37])
def test_mobilenetv3_head_matches_oracle(B, mode):
    from wakeword_trainer_home_amd.models.heads import MobileNetV3Head
    from oracle.mlp_head import MLPHeadOracle
    torch.manual_seed(B)
    head = MobileNetV3Head(576, 1024, 2, dropout=0.3, mode=mode, dropout_seed=5).to(DEV)
    assert list(head.state_dict().keys()) == ["0.weight", "0.bias", "3.weight", "3.bias"]      # the reference's classifier keys
    oracle = MLPHeadOracle(576, 1024, 2, dropout=0.3, seed=5)
    oracle.classifier.load_state_dict({k: v.cpu().double() for k, v in head.state_dict().items()})
    x = torch.randn(B, 576)
    y = torch.randint(0, 2, (B,))
    head.train()
    head[0].sample_offset = 11
    for step in range(2):                                   # the dropout stream advances per training forward
        xd = x.to(DEV).requires_grad_(True)
        out = head(xd)
        loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
        head.zero_grad()
        loss.backward()
        xo = x.double().requires_grad_(True)
        ref = oracle(xo, step=step, sample_offset=11, training=True, bf16=(mode == "bf16"))
        lo = torch.nn.functional.cross_entropy(ref, y)
        oracle.zero_grad()
        lo.backward()
        tol = 5e-6 if mode == "fp32" else 2e-3        # bf16: h is re-rounded on the device from fp32, in the oracle from float64
        assert _rel(out.detach().cpu().double(), ref.detach()) <= tol
        assert abs(loss.item() - lo.item()) <= tol
        for (n, p), q in zip(head.named_parameters(), oracle.classifier.parameters()):
            assert _rel(p.grad.cpu().double(), q.grad) <= (2e-5 if mode == "fp32" else 2e-2), (step, n)
        assert _rel(xd.grad.cpu().double(), xo.grad) <= (2e-5 if mode == "fp32" else 2e-2)
    head.eval()                                             # eval: no dropout
    with torch.no_grad():
        ev = head(x.to(DEV))
    assert _rel(ev.cpu().double(), oracle(x.double(), training=False, bf16=(mode == "bf16")).detach()) <= (5e-6 if mode == "fp32" else 2e-3)


def test_linear_argument_checks():
    from wakeword_trainer_home_amd import _native as nat
    from wakeword_trainer_home_amd.models.heads import MFMALinear
    x = torch.zeros(4, 8, device=DEV)
    with pytest.raises(ValueError):
        nat.linear_mfma_fwd(x, torch.zeros(3, 9, device=DEV))
    with pytest.raises(ValueError):
        nat.linear_mfma_fwd(x, torch.zeros(3, 8, device=DEV), dropout_p=1.0)
    with pytest.raises(ValueError):
        nat.linear_mfma_fwd(x, torch.zeros(3, 8, device=DEV), act=9)
    with pytest.raises(ValueError):
        nat.linear_mfma_bwd(x, torch.zeros(3, 8, device=DEV), None, torch.zeros(4, 3, device=DEV), act=nat.LIN_HARDSWISH)
    with pytest.raises(ValueError):
        MFMALinear(8, 3, activation="relu")
    with pytest.raises(nat.NativeError):
        MFMALinear(8, 3)(torch.zeros(4, 8))


def test_deferred_partial_sums_equal_the_immediate_ones():
    """ww_ctx_set_deferred_reduce: the weight-gradient calls queue their "sum the partials" step and ONE ww_deferred_reduce_flush
    launch runs all of them -- several items of different sizes / split counts in one batch, against the float64 products and
    against the immediate (per-call) sums of the same partials.  Shapes: a split-K dW (M = 30720 rows, 60 splits), a small one
    (no split: nothing is queued), a depthwise 5x5 weight gradient and the stem's."""
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(3)
    lib, cx = nat.load(), nat.ctx(DEV)
    jobs = []
    for M, K, N in ((30720, 96, 576), (7680, 576, 96), (40, 16, 8)):
        x, w, dy = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), torch.randn(M, N, generator=g).to(DEV)
        jobs.append(("pw", x, w, dy))
    xdw, wdw = torch.randn(32, 5, 19, 96, generator=g).to(DEV), torch.randn(96, 1, 5, 5, generator=g).to(DEV)
    dydw = torch.randn(32, 5, 19, 96, generator=g).to(DEV)
    xs, dys = torch.randn(16, 40, 151, generator=g).to(DEV), torch.randn(16, 20, 76, 16, generator=g).to(DEV)
    now = [nat.linear_mfma_bwd(x, w, None, dy, mode=torch.bfloat16, need_db=False)[1] for _, x, w, dy in jobs]
    now.append(nat.dwconv_nhwc_bwd(xdw, wdw, dydw, 5, 1)[1])
    now.append(nat.stem3x3s2_bwd_dw(xs, dys, (16, 1, 3, 3)))
    assert lib.ww_deferred_reduce_pending(cx) == 0
    later = [nat.linear_mfma_bwd(x, w, None, dy, mode=torch.bfloat16, need_db=False, defer=True)[1] for _, x, w, dy in jobs]
    later.append(nat.dwconv_nhwc_bwd(xdw, wdw, dydw, 5, 1, defer=True)[1])
    later.append(nat.stem3x3s2_bwd_dw(xs, dys, (16, 1, 3, 3), defer=True))
    assert lib.ww_deferred_reduce_pending(cx) == 4                      # the (40,16,8) product has no split to defer
    nat.deferred_flush(DEV)
    assert lib.ww_deferred_reduce_pending(cx) == 0
    torch.cuda.synchronize()
    for a, b in zip(now, later):
        # same partials; the flush sums them in double, the immediate kernels in float / two-level double: round-off apart
        assert _rel(b.double(), a.double()) <= 2e-6
    r = lambda t: t.cpu().bfloat16().double()
    for (_, x, w, dy), dw in zip(jobs, later):
        assert _rel(dw.cpu().double(), r(dy).t() @ r(x)) <= 2e-5
    ref = torch.nn.functional.conv2d(xdw.cpu().double().permute(0, 3, 1, 2).reshape(1, 32 * 96, 5, 19),
                                     dydw.cpu().double().permute(0, 3, 1, 2).reshape(32 * 96, 1, 5, 19), padding=2, groups=32 * 96)
    assert _rel(later[3].cpu().double().reshape(96, 5, 5), ref.reshape(32, 96, 5, 5).sum(0)) <= 2e-5
