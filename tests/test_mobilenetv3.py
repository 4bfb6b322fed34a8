"""MobileNetV3-small body on the native channels-last layer library against torch.nn in float64:
per-layer (BatchNorm+activation, depthwise k x k / stride, squeeze-excitation pieces, stem patches) and the whole
``MobileNetV3Wakeword`` against ``oracle/mobilenetv3.py`` (torchvision's published architecture restated in torch.nn;
torchvision itself is not installed -> parity with it is unpinned, see the oracle's header).
fp32 device vs float64: layers <= 2e-5 relative, whole model logits <= 2e-4, gradients <= 2e-3 of the vector norm."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


ACTS = {0: lambda z: z, 1: Fn.hardswish, 2: torch.relu, 3: Fn.hardsigmoid}


@pytest.mark.parametrize("M,C,act,training", [(1000, 16, 1, True), (333, 72, 2, True), (64, 576, 1, True), (50, 24, 0, True),
                                              (200, 40, 3, True), (300, 88, 2, False)])
def test_bn_act_layer(M, C, act, training):
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g, dtype=torch.float64) * 2 + 1).requires_grad_(True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g, dtype=torch.float64) * 0.1, torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    da = torch.randn(M, C, generator=g, dtype=torch.float64)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = Fn.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=training, momentum=0.01, eps=1e-3)
    y_ref = ACTS[act](z)
    (y_ref * da).sum().backward()
    f = lambda t: t.detach().float().to(DEV).contiguous()
    rmd, rvd = f(rm), f(rv)
    gm, bt = f(gamma), f(beta)
    bn = nat.make_bn(gm, bt, rmd, rvd, momentum=0.01, eps=1e-3, training=training)
    xd = f(x)
    y, ss, mr = nat.bn_act_fwd(xd, bn, act, C)
    assert _rel(y.cpu().double(), y_ref.detach()) <= 2e-5
    assert _rel(rmd.cpu().double(), rm_ref) <= 1e-5 and _rel(rvd.cpu().double(), rv_ref) <= 1e-5
    dx, dg, db = nat.bn_act_bwd(xd, f(da), ss, mr, act, training, C)
    assert _rel(dx.cpu().double(), x.grad) <= 5e-5
    assert _rel(dg.cpu().double(), gamma.grad) <= 5e-5 and _rel(db.cpu().double(), beta.grad) <= 5e-5


@pytest.mark.parametrize("B,H,W,C,k,s", [(3, 20, 76, 16, 3, 2), (2, 10, 38, 72, 3, 2), (2, 5, 19, 96, 5, 2), (2, 3, 10, 240, 5, 1),
                                         (1, 2, 5, 576, 5, 1), (2, 7, 9, 10, 3, 1), (5, 5, 19, 88, 3, 1), (9, 3, 10, 288, 5, 2),
                                         (21, 2, 5, 24, 5, 1), (3, 11, 11, 40, 3, 2)])
def test_depthwise_layer(B, H, W, C, k, s):
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(C, 1, k, k, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    y_ref = Fn.conv2d(x, w, stride=s, padding=k // 2, groups=C)
    dy = torch.randn_like(y_ref)
    (y_ref * dy).sum().backward()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).float().contiguous().to(DEV)
    y = nat.dwconv_nhwc_fwd(nhwc(x), w.detach().float().to(DEV), k, s)
    assert y.shape == nhwc(y_ref).shape and _rel(y.cpu().double(), nhwc(y_ref).cpu().double()) <= 2e-6
    dx, dw = nat.dwconv_nhwc_bwd(nhwc(x), w.detach().float().to(DEV), nhwc(dy), k, s)
    assert _rel(dx.cpu().double(), nhwc(x.grad).cpu().double()) <= 2e-6
    assert _rel(dw.cpu().double(), w.grad) <= 2e-5


def test_se_pieces_and_stem_patches():
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(5)
    B, HW, C = 6, 50, 72
    x = torch.randn(B, HW, C, generator=g)
    gate = torch.rand(B, C, generator=g)
    dy = torch.randn(B, HW, C, generator=g)
    dpool = torch.randn(B, C, generator=g)
    xd, gd, dyd, dpd = (t.to(DEV) for t in (x, gate, dy, dpool))
    assert _rel(nat.pool_hw_fwd(xd).cpu(), x.mean(1)) <= 1e-6
    assert torch.equal(nat.scale_bc_fwd(xd, gd).cpu(), x * gate[:, None, :])
    assert _rel(nat.scale_bc_bwd_gate(xd, dyd).cpu(), (x * dy).sum(1)) <= 1e-6
    assert _rel(nat.scale_pool_bwd(dyd, gd, dpd, (B, HW, C)).cpu(), dy * gate[:, None, :] + dpool[:, None, :] / HW) <= 1e-6
    assert _rel(nat.scale_pool_bwd(None, None, dpd, (B, HW, C)).cpu(), (dpool[:, None, :] / HW).expand(B, HW, C)) <= 1e-6
    assert torch.equal(nat.add_f32(xd, dyd).cpu(), x + dy)
    img = torch.randn(3, 1, 41, 151, generator=g)
    cols = nat.im2col3x3s2(img[:, 0].contiguous().to(DEV)).cpu()
    ref = Fn.unfold(img, 3, padding=1, stride=2).transpose(1, 2).reshape(-1, 9)
    assert torch.equal(cols, ref)


@pytest.mark.parametrize("M,K,N,act", [(7680, 40, 240, 1), (2560, 96, 576, 1), (1000, 16, 72, 2), (24320, 72, 24, 0), (300, 9, 16, 1),
                                        (97280, 16, 16, 0)])
def test_conv1x1_bn_act_statistics_in_the_gemm_epilogue(M, K, N, act):
    """ww_conv1x1_bn_act_fwd (GEMM whose epilogue leaves the BatchNorm partials, apply pass that finishes them) against float64
    torch and against the four-launch composition of the same layer; both the finishing apply pass (few row tiles) and the
    finish-launch form (many) are covered by the row counts."""
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g, dtype=torch.float64)
    w = torch.randn(N, K, generator=g, dtype=torch.float64) * 0.3
    gamma, beta = torch.rand(N, generator=g, dtype=torch.float64) + 0.5, torch.randn(N, generator=g, dtype=torch.float64) * 0.3
    rm, rv = torch.randn(N, generator=g, dtype=torch.float64) * 0.1, torch.rand(N, generator=g, dtype=torch.float64) + 0.5
    y_ref = x @ w.T
    rm_ref, rv_ref = rm.clone(), rv.clone()
    a_ref = ACTS[act](Fn.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.01, eps=1e-3))
    f = lambda t: t.float().to(DEV).contiguous()
    xd, wd, gm, bt = f(x), f(w), f(gamma), f(beta)
    rm1, rv1, rm2, rv2 = f(rm), f(rv), f(rm), f(rv)
    y, a, ss, mr = nat.conv1x1_bn_act_fwd(xd, wd, nat.make_bn(gm, bt, rm1, rv1, momentum=0.01, eps=1e-3, training=True), act)
    assert _rel(y.cpu().double(), y_ref) <= 2e-6 and _rel(a.cpu().double(), a_ref) <= 2e-5
    assert _rel(rm1.cpu().double(), rm_ref) <= 1e-5 and _rel(rv1.cpu().double(), rv_ref) <= 1e-5
    y0 = nat.linear_mfma_fwd(xd, wd, None)
    a0, ss0, mr0 = nat.bn_act_fwd(y0, nat.make_bn(gm, bt, rm2, rv2, momentum=0.01, eps=1e-3, training=True), act, N)
    assert torch.equal(y, y0)                                   # the same GEMM
    assert _rel(ss.cpu(), ss0.cpu()) <= 2e-6 and _rel(mr.cpu(), mr0.cpu()) <= 2e-6 and _rel(a.cpu(), a0.cpu()) <= 5e-6
    rm3, rv3 = f(rm), f(rv)                                     # (make_bn holds raw pointers: the tensors must outlive the call)
    y2, a2, ss2, _ = nat.conv1x1_bn_act_fwd(xd, wd, nat.make_bn(gm, bt, rm3, rv3, momentum=0.01, eps=1e-3, training=True), act)
    assert torch.equal(a, a2) and torch.equal(ss, ss2)          # fixed-order sums
    res = torch.randn(M, N, generator=g).to(DEV)                # the inverted-residual skip connection, added in the same pass
    rm4, rv4 = f(rm), f(rv)
    _, a3, _, _ = nat.conv1x1_bn_act_fwd(xd, wd, nat.make_bn(gm, bt, rm4, rv4, momentum=0.01, eps=1e-3, training=True), act, residual=res)
    assert torch.equal(a3, a + res)


@pytest.mark.parametrize("B,H,W,C,k,s,act", [(9, 3, 10, 240, 5, 1, 1), (5, 5, 19, 96, 5, 2, 1), (6, 2, 5, 576, 5, 1, 1), (3, 5, 19, 88, 3, 1, 2),
                                             (2, 10, 38, 72, 3, 2, 2), (2, 7, 9, 10, 3, 1, 2)])
def test_dwconv_bn_act_statistics_from_the_conv_kernel(B, H, W, C, k, s, act):
    """ww_dwconv_bn_act_fwd against float64 torch (depthwise conv -> train-mode BatchNorm -> activation); the last two shapes take
    the composition inside (more than 128 input pixels / C % 4 != 0)."""
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(C, 1, k, k, generator=g, dtype=torch.float64) * 0.3
    gamma, beta = torch.rand(C, generator=g, dtype=torch.float64) + 0.5, torch.randn(C, generator=g, dtype=torch.float64) * 0.3
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y_ref = Fn.conv2d(x, w, stride=s, padding=k // 2, groups=C)
    a_ref = ACTS[act](Fn.batch_norm(y_ref, rm, rv, gamma, beta, training=True, momentum=0.01, eps=1e-3))
    nhwc = lambda t: t.permute(0, 2, 3, 1).float().contiguous().to(DEV)
    f = lambda t: t.float().to(DEV).contiguous()
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    gm, bt = f(gamma), f(beta)
    y, a, ss, mr = nat.dwconv_bn_act_fwd(nhwc(x), f(w), k, s, nat.make_bn(gm, bt, rmd, rvd, momentum=0.01, eps=1e-3, training=True), act)
    assert _rel(y.cpu().double(), nhwc(y_ref).cpu().double()) <= 2e-6
    assert _rel(a.cpu().double(), nhwc(a_ref).cpu().double()) <= 2e-5
    assert _rel(rmd.cpu().double(), rm) <= 1e-5 and _rel(rvd.cpu().double(), rv) <= 1e-5


@pytest.mark.parametrize("B,H,W,C,act", [(3, 40, 151, 16, 1), (2, 7, 9, 8, 2), (130, 12, 11, 32, 1)])
def test_stem_direct_convolution_with_batchnorm(B, H, W, C, act):
    """ww_stem3x3s2_bn_act_fwd / ww_stem3x3s2_bwd_dw (Conv2d(1, C, 3, 2, 1) as a direct convolution whose kernel leaves the
    BatchNorm partials) against float64 torch: outputs, normalised activations, running statistics, weight gradient."""
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, 1, H, W, generator=g, dtype=torch.float64)
    w = (torch.randn(C, 1, 3, 3, generator=g, dtype=torch.float64) * 0.4).requires_grad_(True)
    gamma, beta = torch.rand(C, generator=g, dtype=torch.float64) + 0.5, torch.randn(C, generator=g, dtype=torch.float64) * 0.3
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y_ref = Fn.conv2d(x, w, stride=2, padding=1)
    a_ref = ACTS[act](Fn.batch_norm(y_ref, rm, rv, gamma, beta, training=True, momentum=0.01, eps=1e-3))
    dy = torch.randn_like(y_ref)
    (y_ref * dy).sum().backward()
    f = lambda t: t.detach().float().to(DEV).contiguous()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).float().contiguous().to(DEV)
    gm, bt, rmd, rvd = f(gamma), f(beta), torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xd = f(x[:, 0])
    y, a, ss, mr = nat.stem3x3s2_bn_act_fwd(xd, f(w), nat.make_bn(gm, bt, rmd, rvd, momentum=0.01, eps=1e-3, training=True), act)
    assert _rel(y.cpu().double(), nhwc(y_ref).cpu().double()) <= 2e-6
    assert _rel(a.cpu().double(), nhwc(a_ref).cpu().double()) <= 2e-5
    assert _rel(rmd.cpu().double(), rm) <= 1e-5 and _rel(rvd.cpu().double(), rv) <= 1e-5
    dw = nat.stem3x3s2_bwd_dw(xd, nhwc(dy), w.shape)
    assert _rel(dw.cpu().double(), w.grad) <= 2e-5
    assert torch.equal(dw, nat.stem3x3s2_bwd_dw(xd, nhwc(dy), w.shape))      # fixed-order sums


@pytest.mark.parametrize("B,HW,C,Cs", [(5, 190, 16, 8), (3, 30, 96, 24), (6, 30, 240, 64), (2, 10, 576, 144), (7, 9, 40, 12),
                                       (130, 6, 24, 8), (41, 4, 32, 8), (129, 10, 576, 144), (34, 3, 1024, 256)])
def test_se_block_one_launch_forward_two_backward(B, HW, C, Cs):
    """ww_se_fwd / ww_se_bwd against the block in float64 torch (torchvision SqueezeExcitation semantics: avgpool -> fc1 ->
    ReLU -> fc2 -> Hardsigmoid -> scale); the batch sizes cover the 1 / 2 / 4 images-per-workgroup variants and ragged tails."""
    from wakeword_trainer_home_amd import _native as nat
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, HW, C, generator=g, dtype=torch.float64).requires_grad_(True)
    w1 = (torch.randn(Cs, C, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    b1 = (torch.randn(Cs, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True)
    w2 = (torch.randn(C, Cs, generator=g, dtype=torch.float64) * 0.5).requires_grad_(True)
    b2 = (torch.randn(C, generator=g, dtype=torch.float64) * 0.5).requires_grad_(True)
    dy = torch.randn(B, HW, C, generator=g, dtype=torch.float64)
    s_ref = x.mean(1)
    pre1_ref = s_ref @ w1.T + b1
    pre2_ref = torch.relu(pre1_ref) @ w2.T + b2
    y_ref = x * Fn.hardsigmoid(pre2_ref)[:, None, :]
    (y_ref * dy).sum().backward()
    f = lambda t: t.detach().float().to(DEV).contiguous()
    xd, w1d, b1d, w2d, b2d = f(x), f(w1), f(b1), f(w2), f(b2)
    y, s, pre1, pre2 = nat.se_fwd(xd, w1d, b1d, w2d, b2d)
    assert _rel(s.cpu().double(), s_ref.detach()) <= 1e-6
    assert _rel(pre1.cpu().double(), pre1_ref.detach()) <= 5e-6 and _rel(pre2.cpu().double(), pre2_ref.detach()) <= 5e-6
    assert _rel(y.cpu().double(), y_ref.detach()) <= 5e-6
    dx, dw1, db1, dw2, db2 = nat.se_bwd(xd, f(dy), s, pre1, pre2, w1d, w2d)
    assert _rel(dx.cpu().double(), x.grad) <= 1e-5
    for got, want in ((dw1, w1.grad), (db1, b1.grad), (dw2, w2.grad), (db2, b2.grad)):
        assert _rel(got.cpu().double(), want) <= 2e-5
    y2, *_ = nat.se_fwd(xd, w1d, b1d, w2d, b2d)                 # fixed-order sums: the same bits on every launch
    dx2, dw1b, *_ = nat.se_bwd(xd, f(dy), s, pre1, pre2, w1d, w2d)
    assert torch.equal(y, y2) and torch.equal(dx, dx2) and torch.equal(dw1, dw1b)


def test_mobilenetv3_matches_oracle():
    from wakeword_trainer_home_amd.models import create_model
    from oracle.mobilenetv3 import MobileNetV3Oracle
    from tests.golden_util import make_inputs
    torch.manual_seed(11)
    model = create_model("mobilenetv3", num_classes=2, dropout=0.3, dropout_seed=6).to(DEV)
    oracle = MobileNetV3Oracle(2, dropout=0.3, seed=6)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    assert sum(p.numel() for p in model.parameters()) == 1519618          # mobilenet_v3_small body (1 input channel) + the reference head
    oracle.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in model.state_dict().items()})
    x, y = make_inputs(3, 10)
    model.train()
    oracle.train()
    out = model(x.to(DEV))
    loss = Fn.cross_entropy(out, y.to(DEV))
    loss.backward()
    ref = oracle(x, step=0, training=True)
    lo = Fn.cross_entropy(ref, y)
    lo.backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 2e-4
    assert abs(loss.item() - lo.item()) <= 2e-4
    gd = torch.cat([p.grad.flatten().cpu().double() for p in model.parameters()])
    go = torch.cat([p.grad.flatten() for p in oracle.parameters()])
    assert ((gd - go).norm() / go.norm()).item() <= 2e-3
    for (n, p), q in zip(model.named_parameters(), oracle.parameters()):        # and no tensor is off by itself
        assert (p.grad.cpu().double() - q.grad).norm().item() <= 1e-2 * q.grad.norm().item() + 1e-7, n
    model.state_dict()                      # folds the host-side num_batches_tracked counts into the buffers
    for (n, b), (_, c) in zip(model.named_buffers(), oracle.named_buffers()):   # running statistics after one training step
        assert (b.cpu().double() - c.double()).abs().max().item() <= 1e-4 * c.double().abs().max().item() + 1e-7, n
    model.eval()
    oracle.eval()
    with torch.no_grad():
        ev = model(x.to(DEV))
        ev_ref = oracle(x, training=False)
    assert (ev.cpu().double() - ev_ref).abs().max().item() <= 2e-4


def test_reference_architecture_smoke_shapes():
    """The reference's own model test (tests/test_training_pipeline.py:49-82): every architecture built by ``create_model``
    maps its test input to (4, 2) -- (4,1,64,50) spectrograms for mobilenetv3, (4,50,40) sequences for gru."""
    from wakeword_trainer_home_amd.models import create_model
    for arch, shape in (("mobilenetv3", (4, 1, 64, 50)), ("gru", (4, 50, 40)), ("cnn_small", (4, 1, 64, 50)), ("crnn", (4, 1, 64, 50))):
        model = create_model(arch, num_classes=2, pretrained=False).to(DEV)
        out = model(torch.randn(*shape, device=DEV))
        assert out.shape == (4, 2) and torch.isfinite(out).all(), arch
        out.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters()), arch


def test_trainer_step_uses_flat_buckets_and_fused_optimizer(tmp_path):
    """MobileNetV3 through the Trainer: parameters live in one flat bucket (state_dict keys unchanged), the optimizer is the
    fused clip+update kernel, and one training step equals clip_grad_norm_ + torch.optim.AdamW applied to a copy of the
    model fed the same gradients."""
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    from wakeword_trainer_home_amd.training.optimizer_factory import FlatFusedOptimizer
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size, cfg.optimizer.warmup_epochs = 4, 0
    torch.manual_seed(3)
    model = create_model("mobilenetv3", dropout=0.0)
    keys = list(model.state_dict().keys())
    tr = Trainer(model, [], [], cfg, checkpoint_dir=tmp_path, device=DEV)
    assert isinstance(tr.optimizer, FlatFusedOptimizer) and tr._async_autograd
    assert list(model.state_dict().keys()) == keys and len(keys) == 244
    assert model.flat_param.numel() == sum(p.numel() for p in model.parameters()) == 1519618
    before = [p.detach().clone() for p in model.parameters()]
    x = torch.randn(4, 1, 40, 151, generator=torch.Generator().manual_seed(1)).to(DEV)
    y = torch.tensor([0, 1, 1, 0])
    model.train()
    import copy
    twin = copy.deepcopy(model)                       # the same step's UNCLIPPED gradients, from a copy that takes no optimizer step
    tr.criterion(twin(x), y.to(DEV)).backward()
    unclipped = torch.cat([p.grad.flatten() for p in twin.parameters()]).norm().item()
    tr._step_autograd_async(x, y, 0)
    done = tr._flush_pending()
    assert len(done) == 1 and np.isfinite(done[0][1])
    assert abs(tr.last_grad_norm - unclipped) <= 1e-5 * unclipped
    # the backward kernels wrote the gradients into the flat bucket (p.grad are views of it) and the fused step clipped them THERE,
    # as clip_grad_norm_ leaves p.grad: what is read back is the clipped gradient, the reported norm is the one before clipping
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(model._fb_plist, model._fb_views))
    grads = [p.grad.detach().clone() for p in model.parameters()]
    clip = float(tr.gradient_clip)
    total = torch.cat([g.flatten() for g in grads]).norm().item()
    assert tr.last_grad_norm > clip and abs(total - clip) <= 1e-4 * clip
    ref = [torch.nn.Parameter(b.clone()) for b in before]
    for r, g in zip(ref, grads):
        r.grad = g.clone()
    topt = torch.optim.AdamW(ref, lr=tr.optimizer.param_groups[0]["lr"], betas=tr.optimizer.param_groups[0]["betas"],
                             weight_decay=tr.optimizer.param_groups[0]["weight_decay"])
    topt.step()
    for p, r in zip(model.parameters(), ref):
        assert (p.detach() - r.detach()).abs().max().item() <= 5e-6


def test_backward_that_raises_midway_leaves_no_stale_deferred_sums():
    """The weight-gradient partial sums of a backward pass are queued and flushed by an autograd-engine callback at the END of
    the pass.  A pass that raises midway never reaches it: the next forward must forget the stale queue (its buffers are gone)
    and the next backward must arm a fresh flush -- the gradients of the step after the failure equal those of a model that
    never failed, bit for bit."""
    from tests.golden_util import make_inputs
    from wakeword_trainer_home_amd import _native as nat
    from wakeword_trainer_home_amd.models import create_model
    x, y = make_inputs(12, 8)
    x, y = x.to(DEV), y.to(DEV)

    def grads(model):
        model.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(model(x), y).backward()
        return [p.grad.detach().clone() for p in model.parameters()]

    torch.manual_seed(3)
    clean = create_model("mobilenetv3", dropout=0.0).to(DEV).train()
    torch.manual_seed(3)
    hurt = create_model("mobilenetv3", dropout=0.0).to(DEV).train()
    ref = grads(clean)

    class Boom(RuntimeError):
        pass

    def explode(_):
        raise Boom("backward interrupted")
    hurt.zero_grad(set_to_none=True)
    h = hurt.mobilenet.features[:6](x.float().contiguous())
    h.register_hook(explode)                                    # fires after the late blocks' backward nodes have queued their sums
    from wakeword_trainer_home_amd.models.mobilenet import _PoolFn
    out = hurt.mobilenet.classifier(_PoolFn.apply(hurt.mobilenet.features[6:](h)))
    with pytest.raises(Boom):
        torch.nn.functional.cross_entropy(out, y).backward()
    # (this PyTorch runs the engine's queued callbacks even when the pass raises, so the queue may already be empty here; the
    #  reset at the next forward covers the versions / failure modes that do not)
    hurt.load_state_dict(clean.state_dict())                    # (BatchNorm running statistics advanced in the failed forward)
    clean2 = grads(clean)
    got = grads(hurt)
    assert nat.load().ww_deferred_reduce_pending(nat.ctx(DEV)) == 0
    for a, b, c in zip(got, clean2, ref):
        assert torch.equal(a, b)
