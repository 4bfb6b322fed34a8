"""BASELINE-sized runs (512 clips x 24000 samples; 4096-row recurrent batches) checked through properties that do not need
the oracle at full size: bit-exact index tables, batch-invariance (a clip's result does not depend on what else is in the
batch or where it sits in it), a subset against the oracle, and one full-size training step against the fp32 CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, N = 512, 24000
SA = dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2, freq_mask_prob=0.5, time_mask_prob=0.5)


def test_front_end_full_batch():
    from wakeword_trainer_home_amd import _native as nat
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    from oracle import features as of
    from oracle.specaugment import specaug_indices, specaug_apply
    wave, _ = make_synthetic_batch(B, N, seed=3, device=DEV)
    feat, idx = nat.logmel_fwd(wave, nat.make_feat_cfg(), nat.make_specaug_cfg(**SA), seed=7, step=5, sample_offset=1000,
                               want_idx=True)
    assert feat.shape == (B, 1, 40, 151) and torch.isfinite(feat).all()
    ref_idx = specaug_indices(B, 40, 151, seed=7, step=5, sample_offset=1000, **SA)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)                         # all 512 mask tables, bit-exact
    sub = np.arange(0, B, 37)                                                 # 14 clips against the float64 oracle
    ref = specaug_apply(of.logmel(wave[sub].cpu().numpy()), ref_idx[sub], 2)
    assert np.abs(feat[sub].cpu().numpy() - ref).max() <= 1e-3
    # batch invariance: the same clips alone, addressed by their global sample index, give the same bits
    for i in (0, 255, 511):
        one = nat.logmel_fwd(wave[i:i + 1].contiguous(), nat.make_feat_cfg(), nat.make_specaug_cfg(**SA), seed=7, step=5,
                             sample_offset=1000 + i)
        assert torch.equal(one[0], feat[i])
    # int16 PCM input of the same clips: the quantisation error only
    pcm = (wave * 32767).round().clamp(-32768, 32767).to(torch.int16)
    f16 = nat.logmel_fwd(pcm, nat.make_feat_cfg(), None)
    f32 = nat.logmel_fwd(pcm.float() / 32768.0, nat.make_feat_cfg(), None)
    assert (f16 - f32).abs().max().item() <= 1e-4


@pytest.mark.parametrize("act", ["fp32", "bf16"])
def test_cnn_small_eval_is_batch_invariant_at_full_batch(act):
    """eval mode (running statistics): logits of clip i do not depend on the batch it is evaluated in -- bit for bit."""
    from wakeword_trainer_home_amd.models import create_model
    torch.manual_seed(0)
    model = create_model("cnn_small", dropout=0.3, act_dtype=act).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, 1, 40, 151, device=DEV, generator=g) * 2 - 4
    model.train()
    model(x[:64])                                   # move the running statistics off their initial values
    model.eval()
    with torch.no_grad():
        full = model(x)
        parts = torch.cat([model(x[i:i + 100].contiguous()) for i in range(0, B, 100)])
    assert torch.isfinite(full).all() and torch.equal(full, parts)


def test_full_batch_training_step_matches_oracle():
    """One optimisation step at BASELINE config 2's batch (512 feature maps), fp32 storage, against the plain-torch oracle
    step on the CPU: loss and gradient norm within 1e-3, every parameter after the SGD-Nesterov update within 2e-5 (SGD, not
    Adam: Adam's first step moves every weight by lr * sign(g), which turns round-off in near-zero gradients into lr-sized
    differences -- a property of the optimizer, not of the kernels)."""
    from wakeword_trainer_home_amd.models import create_model, create_loss_function
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss, train_step
    from tests.golden_util import make_inputs
    torch.manual_seed(4)
    model = create_model("cnn_small", dropout=0.3, dropout_seed=9).to(DEV)
    oracle = CNNSmallOracle(dropout=0.3, dropout_seed=9)
    oracle.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    x, y = make_inputs(21, B)
    model.train()
    oracle.train()
    opt = create_optimizer(model, "sgd", learning_rate=0.05, weight_decay=1e-2, momentum=0.9)
    crit = create_loss_function("cross_entropy", label_smoothing=0.05, device=DEV)
    opt.zero_grad(set_to_none=True)
    stats = model.train_step_native(x.to(DEV), y.to(DEV), crit)
    opt.step(max_norm=1.0, stats=stats)
    from wakeword_trainer_home_amd import _native as nat
    s = nat.decode_stats(stats.cpu())
    oopt = torch.optim.SGD(oracle.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2, nesterov=True)
    r = train_step(oracle, TorchLoss("cross_entropy", eps=0.05), oopt, x, y, 1.0)
    assert abs(s["loss"] - r["loss"]) <= 1e-3, (s["loss"], r["loss"])
    assert abs(s["grad_norm"] - r["grad_norm"]) <= 1e-3 * max(r["grad_norm"], 1.0)
    assert abs(s["correct"] / B - r["acc"]) <= 1.0 / B + 1e-9 and s["count"] == B
    for (n, p), q in zip(model.named_parameters(), oracle.parameters()):
        assert (p.detach().cpu() - q.detach()).abs().max().item() <= 2e-5, n


def test_gru_rows_are_independent_at_full_batch():
    """4096 sequences through one GRU direction: each row's output equals the same row run in a batch of 48 (the recurrent
    kernel's 16-row tiles and the GEMM's 64-row tiles place it differently) -- bit for bit."""
    from wakeword_trainer_home_amd import _native as nat
    Bq, T, I, H = 4096, 76, 64, 128
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(Bq, T, I, device=DEV, generator=g)
    w_ih = torch.randn(3 * H, I, device=DEV, generator=g) * 0.1
    w_hh = torch.randn(3 * H, H, device=DEV, generator=g) * 0.1
    b_ih = torch.randn(3 * H, device=DEV, generator=g) * 0.1
    b_hh = torch.randn(3 * H, device=DEV, generator=g) * 0.1
    y = torch.empty(Bq, T, H, device=DEV)
    h_n = nat.gru_fwd(x, w_ih, w_hh, b_ih, b_hh, y, nat.gru_workspace(Bq, T, I, H, DEV), reverse=True)
    assert torch.isfinite(y).all() and y.abs().max().item() <= 1.0      # |h| <= 1 by construction of the cell
    rows = slice(1000, 1048)
    y2 = torch.empty(48, T, H, device=DEV)
    h2 = nat.gru_fwd(x[rows].contiguous(), w_ih, w_hh, b_ih, b_hh, y2, nat.gru_workspace(48, T, I, H, DEV), reverse=True)
    assert torch.equal(y[rows], y2) and torch.equal(h_n[rows], h2)
    assert torch.equal(h_n, y[:, 0, :])                                 # reverse direction: the final state is the t = 0 output
