"""Fused clip + optimizer step against torch.optim's own CPU implementations -- the update rules
the reference instantiates in create_optimizer (src/training/optimizer_factory.py:165-199) -- and against
torch.nn.utils.clip_grad_norm_ (:446-452).  fp32 both sides; tolerance 2e-6 abs on parameters of O(1) (a few ulp:
torch's CPU kernels contract differently), state and step counts exact in structure."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_opt(kind, params, lr, wd):
    if kind == "adam":
        return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    if kind == "adamw":
        return torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), weight_decay=wd)
    return torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=wd, nesterov=True)


def _model_pair(seed=0):
    from wakeword_trainer_home_amd.models import create_model
    from oracle.cnn_small import CNNSmallOracle
    torch.manual_seed(seed)
    model = create_model("cnn_small", dropout=0.0).to(DEV)
    ref = CNNSmallOracle(dropout=0.0)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    return model, ref


def _set_grads(model, ref, g):
    """Same gradient values on both sides; on the device they go where backward puts them (the flat bucket)."""
    model._prepare(torch.device(DEV))
    model.flat_grad.copy_(g.to(DEV))
    off = 0
    for p, q in zip(model.parameters(), ref.parameters()):
        n = p.numel()
        p.grad = model._grad_views[id(p)]
        q.grad = g[off:off + n].view_as(q).clone()
        off += n


@pytest.mark.parametrize("kind", ["adam", "adamw", "sgd"])
@pytest.mark.parametrize("max_norm", [0.0, 1.0])
def test_matches_torch_optim(kind, max_norm):
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer, FlatFusedOptimizer
    model, ref = _model_pair()
    opt = create_optimizer(model, kind, learning_rate=3e-3, weight_decay=1e-2, momentum=0.9)
    assert isinstance(opt, FlatFusedOptimizer)
    assert all(p.data_ptr() >= model.flat_param.data_ptr() for p in model.parameters())       # views of the bucket
    topt = _torch_opt(kind, list(ref.parameters()), 3e-3, 1e-2)
    n = model.flat_param.numel()
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        grads = torch.randn(n, generator=g) * (3.0 if step % 2 else 0.01)      # both sides of the clip threshold
        _set_grads(model, ref, grads)
        if step == 3:
            for gp in opt.param_groups:                                        # a scheduler changes lr mid-run
                gp["lr"] = 1e-3
            for gp in topt.param_groups:
                gp["lr"] = 1e-3
        opt.step(max_norm=max_norm)
        if max_norm > 0:
            tn = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
            assert abs(opt.grad_norm.item() - tn.item()) <= 1e-5 * tn.item()
        topt.step()
        for (name, p), q in zip(model.named_parameters(), ref.parameters()):
            assert (p.detach().cpu() - q.detach()).abs().max().item() <= 2e-6, (step, name)
    assert opt.step_count() == 6


def test_nonfinite_step_is_skipped_and_not_counted():
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer
    from wakeword_trainer_home_amd import _native as nat
    model, ref = _model_pair(1)
    opt = create_optimizer(model, "adamw", learning_rate=1e-3, weight_decay=1e-2)
    topt = _torch_opt("adamw", list(ref.parameters()), 1e-3, 1e-2)
    n = model.flat_param.numel()
    g = torch.Generator().manual_seed(2)
    stats = torch.zeros(nat.STEP_STATS_BYTES, dtype=torch.uint8, device=DEV)
    for step in range(5):
        grads = torch.randn(n, generator=g)
        stats.zero_()
        if step == 1:
            grads[77] = float("nan")                      # non-finite gradient norm
        if step == 3:
            stats.view(torch.float32)[nat.FOUND_INF_FLOAT_INDEX] = 1.0     # the loss kernel flagged the batch
        before = model.flat_param.clone()
        _set_grads(model, ref, grads)
        opt.step(max_norm=1.0, stats=stats)
        if step in (1, 3):
            assert torch.equal(model.flat_param, before)
            assert stats.view(torch.float32)[nat.FOUND_INF_FLOAT_INDEX].item() == 1.0
        else:
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
            topt.step()
    assert opt.step_count() == 3
    for p, q in zip(model.parameters(), ref.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max().item() <= 2e-6


@pytest.mark.parametrize("kind", ["adamw", "sgd"])
def test_state_dict_is_torch_compatible(kind):
    """A checkpoint written by the fused optimizer resumes a plain torch optimizer and vice versa (the reference's
    checkpoints hold optimizer.state_dict(), src/training/trainer.py:430-446)."""
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer
    model, ref = _model_pair(2)
    opt = create_optimizer(model, kind, learning_rate=2e-3, weight_decay=1e-2, momentum=0.9)
    topt = _torch_opt(kind, list(ref.parameters()), 2e-3, 1e-2)
    assert opt.state_dict()["state"] == {}
    n = model.flat_param.numel()
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        grads = torch.randn(n, generator=g)
        _set_grads(model, ref, grads)
        opt.step()
        topt.step()
    sd, tsd = opt.state_dict(), topt.state_dict()
    assert set(sd["state"].keys()) == set(tsd["state"].keys()) and sd["param_groups"][0]["params"] == tsd["param_groups"][0]["params"]
    for k in tsd["state"]:
        assert set(sd["state"][k].keys()) == set(tsd["state"][k].keys())
        for name, tv in tsd["state"][k].items():
            assert torch.allclose(sd["state"][k][name].cpu().float(), torch.as_tensor(tv).float(), atol=2e-6), (k, name)
    # fused -> torch
    t2 = _torch_opt(kind, list(ref.parameters()), 2e-3, 1e-2)
    t2.load_state_dict({"state": {k: {a: b.cpu() for a, b in v.items()} for k, v in sd["state"].items()},
                        "param_groups": sd["param_groups"]})
    # torch -> fused (fresh model with the same weights)
    model2, _ = _model_pair(2)
    model2.load_state_dict(model.state_dict())
    o2 = create_optimizer(model2, kind, learning_rate=2e-3, weight_decay=1e-2, momentum=0.9)
    o2.load_state_dict(tsd)
    assert o2.step_count() == (3 if kind != "sgd" else 1)
    grads = torch.randn(n, generator=g)
    _set_grads(model2, ref, grads)
    o2.step()
    t2.step()
    for p, q in zip(model2.parameters(), ref.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max().item() <= 2e-6


def test_large_bucket_path_and_argument_checks():
    """n > 32768 takes the two-launch path (block partial sums of squares + a grid-wide clip/update that reduces them in
    every block): same numbers as torch; a non-finite gradient skips the update and leaves the step count alone."""
    from wakeword_trainer_home_amd import _native as nat
    n = (1 << 17) + 12345
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(n, generator=g)
    q = torch.nn.Parameter(p0.clone())
    topt = torch.optim.AdamW([q], lr=1e-3, weight_decay=1e-2)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    cfg = nat.OptimCfg(nat.OPT_ADAMW, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.0, 5.0)
    norm = torch.zeros(1, device=DEV)
    for k in range(3):
        grads = torch.randn(n, generator=g)
        q.grad = grads.clone()
        tn = torch.nn.utils.clip_grad_norm_([q], 5.0)
        topt.step()
        nat.clip_optim_step_(cfg, p, grads.to(DEV), m, v, state, k & 1, norm_out=norm)
        assert abs(norm.item() - tn.item()) <= 1e-5 * tn.item()
        assert (p.cpu() - q.detach()).abs().max().item() <= 2e-6
    assert state[1].item() == 3
    bad = torch.randn(n, generator=g)
    bad[n // 2] = float("inf")
    before = p.clone()
    stats = torch.zeros(nat.STEP_STATS_BYTES, dtype=torch.uint8, device=DEV)
    nat.clip_optim_step_(cfg, p, bad.to(DEV), m, v, state, 1, norm_out=norm, stats=stats)
    assert torch.equal(p, before) and state[0].item() == 3
    assert stats.view(torch.float32)[nat.FOUND_INF_FLOAT_INDEX].item() == 1.0
    # the stand-alone clip entry point on a large bucket (block partials + grid-wide apply)
    gl = torch.randn(n, generator=g) * 3
    gd = gl.to(DEV)
    qq = torch.nn.Parameter(torch.zeros(n))
    qq.grad = gl.clone()
    tn = torch.nn.utils.clip_grad_norm_([qq], 2.0)
    got = nat.grad_norm_clip_(gd, 2.0)
    assert abs(float(got) - tn.item()) <= 1e-5 * tn.item()
    assert (gd.cpu() - qq.grad).abs().max().item() <= 1e-6
    with pytest.raises(ValueError, match="Learning rate must be positive"):
        nat.clip_optim_step_(nat.OptimCfg(nat.OPT_ADAMW, 0.0, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0), p, p.clone(), m, v, state, 0)
    with pytest.raises(ValueError, match="Betas must be in"):
        nat.clip_optim_step_(nat.OptimCfg(nat.OPT_ADAM, 1e-3, 1.5, 0.999, 1e-8, 0.0, 0.0, 0.0), p, p.clone(), m, v, state, 0)
    with pytest.raises(ValueError):
        nat.clip_optim_step_(nat.OptimCfg(7, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0), p, p.clone(), m, v, state, 0)


def test_model_moved_after_optimizer_creation_fails_loudly():
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer
    from wakeword_trainer_home_amd import _native as nat
    model, ref = _model_pair(3)
    opt = create_optimizer(model, "adamw", learning_rate=1e-3)
    model.to("cpu")                       # a real move: every parameter gets storage of its own ...
    model.to(DEV)                         # ... so the model builds a NEW bucket (a no-op .to() keeps the old one: the
                                          # parameters still sit back to back in it and are adopted as they are)
    _set_grads(model, ref, torch.zeros(model.flat_param.numel()))
    with pytest.raises(nat.NativeError, match="moved after the optimizer was created"):
        opt.step()


def test_frozen_parameters_go_to_torch_optim_not_the_fused_kernel():
    """torch.optim skips a parameter without a gradient (no decay, no moment update); the fused kernel walks the whole
    bucket.  So a model with a frozen parameter gets torch.optim from create_optimizer, the fused class refuses it, and a
    bucketed model whose backward left a parameter without a gradient raises instead of updating it with a zero gradient."""
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training.optimizer_factory import FlatFusedOptimizer, create_optimizer
    model = create_model("cnn_small", dropout=0.0).to(DEV)
    model.stem.conv.weight.requires_grad_(False)
    opt = create_optimizer(model, "adamw", learning_rate=1e-3, weight_decay=1e-2)
    assert isinstance(opt, torch.optim.AdamW) and not isinstance(opt, FlatFusedOptimizer)
    with pytest.raises(ValueError, match="frozen"):
        FlatFusedOptimizer(model, "adamw", 1e-3)
    m2 = create_model("mobilenetv3", dropout=0.0).to(DEV)
    o2 = create_optimizer(m2, "adamw", learning_rate=1e-3, weight_decay=1e-2)
    assert isinstance(o2, FlatFusedOptimizer)
    m2.train()
    m2(torch.randn(2, 1, 40, 51, device=DEV)).sum().backward()
    next(iter(m2.parameters())).grad = None
    with pytest.raises(RuntimeError, match="no gradient"):
        m2.gather_grads()
    sgd = create_optimizer(create_model("cnn_small", dropout=0.0).to(DEV), "sgd", learning_rate=0.1, momentum=0.0)
    x = torch.randn(4, 1, 40, 51, device=DEV)
    sgd._model.train()
    sgd._model(x).sum().backward()
    sgd.step()
    assert all(st["momentum_buffer"] is None for st in sgd.state_dict()["state"].values())      # as torch.optim.SGD saves it
