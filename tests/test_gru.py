"""GRU layer kernels against torch.nn.GRU on the CPU in float64 -- the module the reference's
GRUWakeword wraps (src/models/architectures.py:228-235), so its arithmetic IS the reference's.  fp32 device vs float64:
outputs <= 2e-5 abs (76 recurrent steps of fp32 sigmoid/tanh), gradients <= 2e-4 relative to each tensor's largest entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


@pytest.fixture(params=[8, 16])
def gru_rows(request, monkeypatch):
    """Both workgroup shapes of the recurrent kernels (8 or 16 batch rows per workgroup; the library picks by batch size)."""
    monkeypatch.setenv("WW_GRU_ROWS", str(request.param))
    return request.param


@pytest.mark.parametrize("B,T,I", [(5, 7, 40), (33, 76, 64), (16, 3, 9), (70, 20, 256)])
@pytest.mark.parametrize("reverse", [False, True])
def test_gru_direction_matches_torch(B, T, I, reverse, gru_rows):
    from wakeword_trainer_home_amd import _native as nat
    H = 128
    torch.manual_seed(B + T)
    ref = torch.nn.GRU(I, H, num_layers=1, batch_first=True, bidirectional=True).double()
    sfx = "_reverse" if reverse else ""
    w_ih, w_hh = getattr(ref, "weight_ih_l0" + sfx), getattr(ref, "weight_hh_l0" + sfx)
    b_ih, b_hh = getattr(ref, "bias_ih_l0" + sfx), getattr(ref, "bias_hh_l0" + sfx)
    x = torch.randn(B, T, I, dtype=torch.float64, requires_grad=True)
    h0 = torch.randn(2, B, H, dtype=torch.float64) * 0.3
    out, hn = ref(x, h0)
    sl = slice(H, 2 * H) if reverse else slice(0, H)
    dy = torch.randn(B, T, H, dtype=torch.float64)
    dhn = torch.randn(B, H, dtype=torch.float64)
    loss = (out[:, :, sl] * dy).sum() + (hn[1 if reverse else 0] * dhn).sum()
    loss.backward()

    f = lambda t: t.detach().float().to(DEV)
    ybuf = torch.zeros(B, T, 2 * H, device=DEV)                     # the direction writes its half of a (B,T,2H) buffer
    ws = nat.gru_workspace(B, T, I, H, DEV)
    xd = f(x)
    h_n = nat.gru_fwd(xd, f(w_ih), f(w_hh), f(b_ih), f(b_hh), ybuf[:, :, sl], ws, h0=f(h0[1 if reverse else 0]), reverse=reverse)
    assert (ybuf[:, :, sl].cpu().double() - out[:, :, sl].detach()).abs().max().item() <= 2e-5
    assert (h_n.cpu().double() - hn[1 if reverse else 0].detach()).abs().max().item() <= 2e-5
    other = slice(0, H) if reverse else slice(H, 2 * H)
    assert ybuf[:, :, other].abs().max().item() == 0.0             # nothing written outside its half
    dybuf = torch.zeros(B, T, 2 * H, device=DEV)
    dybuf[:, :, sl] = f(dy)
    dx = torch.full((B, T, I), 1.0, device=DEV)
    dw_ih, dw_hh, db_ih, db_hh, dh0 = nat.gru_bwd(xd, f(w_ih), f(w_hh), dybuf[:, :, sl], f(dhn), ws, reverse=reverse, dx=dx,
                                                  accumulate_dx=True, want_dh0=True)
    tol = 2e-4
    assert _rel(dx.cpu().double() - 1.0, x.grad) <= tol                  # accumulated onto the ones
    assert _rel(dw_ih.cpu().double(), w_ih.grad) <= tol
    assert _rel(dw_hh.cpu().double(), w_hh.grad) <= tol
    assert _rel(db_ih.cpu().double(), b_ih.grad) <= tol
    assert _rel(db_hh.cpu().double(), b_hh.grad) <= tol
    # dh0 = d loss / d h0 of this direction
    h0g = h0.clone().requires_grad_(True)
    out2, hn2 = ref(x.detach(), h0g)
    ((out2[:, :, sl] * dy).sum() + (hn2[1 if reverse else 0] * dhn).sum()).backward()
    assert _rel(dh0.cpu().double(), h0g.grad[1 if reverse else 0]) <= tol


@pytest.mark.parametrize("B,T,I", [(5, 7, 40), (33, 76, 64), (70, 20, 256)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gru_bidirectional_layer_in_one_launch(B, T, I, mode, gru_rows):
    """ww_gru_bidir_fwd/bwd (both directions as the two rows of ONE recurrent launch) against float64 torch.nn.GRU
    (bidirectional=True) in the fp32 mode, and BIT-identical to the two per-direction launches it replaces in either mode
    (same kernels, same per-direction arithmetic) -- except dx: one product over both directions' (dGi, W_ih) pairs here, a
    product plus an accumulating one there: the same terms in another order of fp32 additions."""
    from wakeword_trainer_home_amd import _native as nat
    H = 128
    torch.manual_seed(B * T)
    ref = torch.nn.GRU(I, H, num_layers=1, batch_first=True, bidirectional=True).double()
    x = torch.randn(B, T, I, dtype=torch.float64, requires_grad=True)
    out, hn = ref(x)
    dy = torch.randn(B, T, 2 * H, dtype=torch.float64)
    dhn = torch.randn(2, B, H, dtype=torch.float64)
    ((out * dy).sum() + (hn * dhn).sum()).backward()
    f = lambda t: t.detach().float().to(DEV)
    m = {"fp32": torch.float32, "bf16": torch.bfloat16}[mode]
    names = [["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"], ["weight_ih_l0_reverse", "weight_hh_l0_reverse",
                                                                            "bias_ih_l0_reverse", "bias_hh_l0_reverse"]]
    params = [[f(getattr(ref, n)) for n in ns] for ns in names]
    xd, dyd = f(x), f(dy)
    ws = [nat.gru_workspace(B, T, I, H, DEV) for _ in range(2)]
    y = torch.zeros(B, T, 2 * H, device=DEV)
    h_n = nat.gru_bidir_fwd(xd, params, y, ws, mode=m)
    dx = torch.zeros(B, T, I, device=DEV)
    grads = nat.gru_bidir_bwd(xd, params, dyd, [f(dhn[0]), f(dhn[1])], ws, dx=dx, mode=m)
    # the per-direction launches
    ws1 = [nat.gru_workspace(B, T, I, H, DEV) for _ in range(2)]
    y1 = torch.zeros(B, T, 2 * H, device=DEV)
    h1 = [nat.gru_fwd(xd, *params[d], y1[:, :, d * H:(d + 1) * H], ws1[d], reverse=(d == 1), mode=m) for d in range(2)]
    dx1 = torch.zeros(B, T, I, device=DEV)
    g1 = [nat.gru_bwd(xd, params[d][0], params[d][1], dyd[:, :, d * H:(d + 1) * H], f(dhn[d]), ws1[d], reverse=(d == 1), dx=dx1,
                      accumulate_dx=(d == 1), mode=m) for d in range(2)]
    assert torch.equal(y, y1) and torch.equal(h_n[0], h1[0]) and torch.equal(h_n[1], h1[1])
    ddx = (dx - dx1).abs().max().item() / max(dx1.abs().max().item(), 1e-30)
    print(f"bidirectional dx, one product vs product + accumulate: max |diff| / max |dx| = {ddx:.2e}")
    assert ddx <= 2e-6
    for d in range(2):
        for a, b in zip(grads[d], g1[d][:4]):
            assert torch.equal(a, b)
    if mode == "fp32":
        assert (y.cpu().double() - out.detach()).abs().max().item() <= 2e-5
        assert (torch.stack(h_n).cpu().double() - hn.detach()).abs().max().item() <= 2e-5
        assert _rel(dx.cpu().double(), x.grad) <= 2e-4
        for d in range(2):
            for a, n in zip(grads[d], names[d]):
                assert _rel(a.cpu().double(), getattr(ref, n).grad) <= 2e-4, n


def test_gru_argument_checks():
    from wakeword_trainer_home_amd import _native as nat
    with pytest.raises(nat.NativeError, match="128 only"):
        nat.gru_workspace(4, 5, 8, 64, DEV)
    ws = nat.gru_workspace(4, 5, 8, 128, DEV)
    x = torch.zeros(4, 5, 8, device=DEV)
    w_ih, w_hh = torch.zeros(384, 8, device=DEV), torch.zeros(384, 128, device=DEV)
    b = torch.zeros(384, device=DEV)
    with pytest.raises(ValueError):
        nat.gru_fwd(x, w_ih, w_hh, b, b, torch.zeros(4, 5, 64, device=DEV), ws)
    with pytest.raises(nat.NativeError):
        nat.gru_fwd(x, w_ih, w_hh, b, b, torch.zeros(4, 5, 128, device=DEV), ws[:100])       # workspace too small
    with pytest.raises(ValueError):
        nat.gru_fwd(x.transpose(0, 1), w_ih, w_hh, b, b, torch.zeros(5, 4, 128, device=DEV), ws)


# ------------------------------------------------------------------------------------------ model level
def _golden():
    from pathlib import Path
    return np.load(Path(__file__).parent / "golden" / "g7_gru.npz")


def _sd_from_golden(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}


def test_gruwakeword_matches_reference_fixture():
    """tests/golden/g7_gru.npz was produced by the reference's own GRUWakeword (make_golden.py g7): same state_dict keys,
    eval logits, training loss and every parameter gradient."""
    from wakeword_trainer_home_amd.models import create_model
    g = _golden()
    sd = _sd_from_golden(g)
    model = create_model("gru", num_classes=2, input_size=40, hidden_size=128, num_layers=2, bidirectional=True, dropout=0.0)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.to(DEV)
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    model.eval()
    with torch.no_grad():
        ev = model(x)
    assert (ev.cpu() - torch.from_numpy(g["logits_eval"])).abs().max().item() <= 2e-5
    model.train()
    out = model(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5
    for n, p in model.named_parameters():
        ref = torch.from_numpy(g["grad." + n])
        assert _rel(p.grad.cpu(), ref) <= 5e-4, n
    # (B,1,F,T) feature batches, as the Trainer produces them, give the same logits
    with torch.no_grad():
        model.eval()
        ev4 = model(x.transpose(1, 2)[:, None].contiguous())
    assert torch.equal(ev4, ev)


@pytest.mark.parametrize("layers,bidir", [(2, True), (1, True), (2, False)])
def test_gruwakeword_with_dropout_matches_oracle(layers, bidir):
    from wakeword_trainer_home_amd.models import create_model
    from oracle.gru import GRUWakewordOracle
    torch.manual_seed(layers * 3 + int(bidir))
    model = create_model("gru", input_size=40, num_layers=layers, bidirectional=bidir, dropout=0.3, dropout_seed=21).to(DEV)
    oracle = GRUWakewordOracle(40, 128, layers, 2, bidir, dropout=0.3, seed=21)
    oracle.load_reference_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    B, T = 9, 23
    x = torch.randn(B, T, 40)
    y = torch.randint(0, 2, (B,))
    model.train()
    model.sample_offset = 5
    for step in range(2):
        xd = x.to(DEV).requires_grad_(True)
        out = model(xd)
        loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
        model.zero_grad()
        loss.backward()
        xo = x.double().requires_grad_(True)
        ref = oracle(xo, step=step, sample_offset=5, training=True)
        lo = torch.nn.functional.cross_entropy(ref, y)
        oracle.zero_grad()
        lo.backward()
        assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 5e-5, step
        assert _rel(xd.grad.cpu().double(), xo.grad) <= 5e-4
        nd = 2 if bidir else 1
        for k in range(layers):
            for sfx in ("", "_reverse")[:nd]:
                for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    gd = getattr(model.gru, f"{name}_l{k}{sfx}").grad.cpu().double()
                    go = getattr(oracle.layers[k], f"{name}_l0{sfx}").grad
                    assert _rel(gd, go) <= 5e-4, (step, name, k, sfx)
        assert _rel(model.fc[1].weight.grad.cpu().double(), oracle.fc.weight.grad) <= 5e-4


@pytest.mark.parametrize("act", ["fp32", "bf16"])
def test_crnn_matches_oracle(act):
    """conv front-end + frequency pooling + GRU against oracle/crnn.py (torch.nn conv stack + nn.GRU, float64)."""
    from wakeword_trainer_home_amd.models import create_model
    from oracle.crnn import CRNNOracle
    from tests.golden_util import make_inputs
    torch.manual_seed(3)
    model = create_model("crnn", dropout=0.3, dropout_seed=4, act_dtype=act).to(DEV)
    keys = list(model.state_dict().keys())
    assert keys[0] == "front.stem.conv.weight" and "rnn.gru.weight_hh_l1_reverse" in keys and keys[-1] == "rnn.fc.1.bias"
    assert not any("classifier" in k for k in keys)
    oracle = CRNNOracle(dropout=0.3, seed=4)
    oracle.load_device_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    x, y = make_inputs(5, 12)
    model.train()
    oracle.train()
    xd = x.to(DEV)
    out = model(xd)
    assert out.shape == (12, 2)
    loss = torch.nn.functional.cross_entropy(out, y.to(DEV))
    loss.backward()
    ref = oracle(x, step=0, training=True)
    lo = torch.nn.functional.cross_entropy(ref, y)
    lo.backward()
    tol = 2e-4 if act == "fp32" else 3e-2
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= tol
    assert abs(loss.item() - lo.item()) <= tol
    gd = torch.cat([p.grad.flatten().cpu().double() for n, p in model.named_parameters() if n.startswith("rnn.")])
    go = torch.cat([p.grad.flatten() for p in oracle.rnn.parameters()])
    assert len(gd) == len(go)
    # whole-vector bounds: at B=12 a single fp32 ReLU decision within round-off of zero moves individual conv gradients
    assert ((gd - go).norm() / go.norm()).item() <= (2e-3 if act == "fp32" else 8e-2)
    fd = torch.cat([p.grad.flatten().cpu().double() for n, p in model.named_parameters() if n.startswith("front.")])
    fo = torch.cat([p.grad.flatten() for n, p in oracle.front.named_parameters() if not n.startswith("classifier")])
    if act == "fp32":
        assert ((fd - fo).norm() / fo.norm()).item() <= 2e-2
    else:      # bf16 storage of 9 activation / 9 gradient tensors at B=12: direction of the conv-stack gradient (cf. test_bf16_mode.py)
        assert (fd @ fo / (fd.norm() * fo.norm())).item() > 0.97
    model.eval()                      # running statistics were updated by ONE training forward on both sides
    oracle.eval()
    with torch.no_grad():
        ev = model(xd)
        ev_ref = oracle(x, training=False)
    assert (ev.cpu().double() - ev_ref).abs().max().item() <= (5e-4 if act == "fp32" else 5e-2)


@pytest.mark.parametrize("arch", ["crnn", "gru"])
def test_trainer_drives_recurrent_models(tmp_path, arch):
    """Trainer.train_epoch / validate_epoch on waveform batches with the recurrent models (generic autograd path: native
    front end -> model -> native loss -> torch AdamW); first-step loss == the oracle pipeline's."""
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    from oracle.train_step import frontend
    from oracle.crnn import CRNNOracle
    from oracle.gru import GRUWakewordOracle
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    cfg.loss.label_smoothing = 0.0
    torch.manual_seed(9)
    model = create_model(arch, dropout=0.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    wave, y = make_synthetic_batch(24, 24000, seed=5)
    y[::3] = 1
    batches = [(wave[i:i + 8], y[i:i + 8], [{"path": "s"}] * 8) for i in range(0, 24, 8)]
    t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path, device=DEV)
    losses = []

    class Rec:
        def on_batch_end(self, batch_idx, loss, acc):
            losses.append(loss)
    t.add_callback(Rec())
    t.train_epoch(0)
    assert len(losses) == 3 and all(np.isfinite(losses))
    a = cfg.augmentation
    spec = dict(freq_mask_param=a.freq_mask_param, time_mask_param=a.time_mask_param, n_freq_masks=a.n_freq_masks,
                n_time_masks=a.n_time_masks, freq_mask_prob=a.freq_mask_prob, time_mask_prob=a.time_mask_prob)
    x0, _ = frontend(batches[0][0].numpy(), spec, seed=a.seed, step=0)
    if arch == "crnn":
        oracle = CRNNOracle(dropout=0.0)
        oracle.load_device_state_dict(sd)
    else:
        oracle = GRUWakewordOracle(40, dropout=0.0)
        oracle.load_reference_state_dict(sd)
    oracle.train()
    ref = torch.nn.functional.cross_entropy(oracle(torch.as_tensor(x0), training=True), batches[0][1]).item()
    assert abs(losses[0] - ref) < 1e-3, (losses[0], ref)
    loss, m = t.validate_epoch(0)
    assert m.total_samples == 8 and np.isfinite(loss)


def test_gruwakeword_bf16_matrix_mode():
    """mode='bf16': the projection / weight-gradient GEMMs take bf16 operands (fp32 accumulation), the recurrence stays fp32.
    Against the float64 oracle: logits within 2e-2, gradient direction cos > 0.999."""
    from wakeword_trainer_home_amd.models import create_model
    from oracle.gru import GRUWakewordOracle
    torch.manual_seed(1)
    model = create_model("gru", input_size=40, dropout=0.0, mode="bf16").to(DEV)
    oracle = GRUWakewordOracle(40, dropout=0.0)
    oracle.load_reference_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    x = torch.randn(33, 50, 40)
    y = torch.randint(0, 2, (33,))
    model.train()
    out = model(x.to(DEV))
    torch.nn.functional.cross_entropy(out, y.to(DEV)).backward()
    ref = oracle(x, training=True)
    torch.nn.functional.cross_entropy(ref, y).backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 2e-2
    gd = torch.cat([p.grad.flatten().cpu().double() for p in model.parameters()])
    go = torch.cat([p.grad.flatten() for p in oracle.parameters()])
    assert (gd @ go / (gd.norm() * go.norm())).item() > 0.999


@pytest.mark.parametrize("reverse", [False, True])
def test_gru_direction_bf16_mode(reverse, gru_rows):
    """mode=bf16: h and W_hh enter the per-step MFMA as bf16 (fp32 accumulation, fp32 state): outputs within 2e-2 of the
    float64 nn.GRU over 76 steps, weight-gradient direction cos > 0.999."""
    from wakeword_trainer_home_amd import _native as nat
    B, T, I, H = 40, 76, 64, 128
    torch.manual_seed(5)
    ref = torch.nn.GRU(I, H, num_layers=1, batch_first=True, bidirectional=True).double()
    sfx = "_reverse" if reverse else ""
    P = [getattr(ref, n + "_l0" + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    x = torch.randn(B, T, I, dtype=torch.float64)
    out, hn = ref(x)
    sl = slice(H, 2 * H) if reverse else slice(0, H)
    dy = torch.randn(B, T, H, dtype=torch.float64)
    (out[:, :, sl] * dy).sum().backward()
    f = lambda t: t.detach().float().to(DEV)
    y = torch.empty(B, T, H, device=DEV)
    ws = nat.gru_workspace(B, T, I, H, DEV)
    nat.gru_fwd(f(x), *[f(p) for p in P], y, ws, reverse=reverse, mode=torch.bfloat16)
    assert (y.cpu().double() - out[:, :, sl].detach()).abs().max().item() <= 2e-2
    dw_ih, dw_hh, db_ih, db_hh, _ = nat.gru_bwd(f(x), f(P[0]), f(P[1]), f(dy), None, ws, reverse=reverse, mode=torch.bfloat16)
    for got, p in ((dw_ih, P[0]), (dw_hh, P[1]), (db_ih, P[2]), (db_hh, P[3])):
        a, b = got.cpu().double().flatten(), p.grad.flatten()
        assert (a @ b / (a.norm() * b.norm())).item() > 0.999


def test_async_autograd_step_skips_nonfinite_batch_on_device(tmp_path):
    """The sync-free step of a HIP-backed autograd model: a batch with a non-finite loss changes no parameter (the fused
    optimizer gets the flag on the device), is not reported to callbacks and still counts in the epoch average's
    denominator (reference quirk Q5, trainer.py:177-179,228)."""
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    torch.manual_seed(2)
    model = create_model("gru", dropout=0.0)
    wave, y = make_synthetic_batch(24, 24000, seed=6)
    bad = wave[8:16].clone()
    bad[3, 100] = float("nan")
    batches = [(wave[:8], y[:8]), (bad, y[8:16]), (wave[16:], y[16:])]
    t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path, device=DEV)
    assert t._async_autograd and t.deferred_metrics
    seen = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: seen.append((i, l))})())
    snaps = []
    orig = t._step_autograd_async

    def spy(inputs, targets, idx):
        snaps.append({k: v.clone() for k, v in model.state_dict().items()})
        return orig(inputs, targets, idx)
    t._step_autograd_async = spy
    avg, _ = t.train_epoch(0)
    after = {k: v.clone() for k, v in model.state_dict().items()}
    assert [i for i, _ in seen] == [0, 2] and all(np.isfinite(l) for _, l in seen)
    changed01 = any(not torch.equal(snaps[0][k], snaps[1][k]) for k in snaps[0])
    same12 = all(torch.equal(snaps[1][k], snaps[2][k]) for k in snaps[1])          # the NaN batch left everything untouched
    changed2 = any(not torch.equal(snaps[2][k], after[k]) for k in after)
    assert changed01 and same12 and changed2
    assert abs(avg - (seen[0][1] + seen[1][1]) / 3) < 1e-6
