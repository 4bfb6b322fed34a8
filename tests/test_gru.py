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


@pytest.mark.parametrize("B,T,I", [(5, 7, 40), (33, 76, 64), (16, 3, 9), (70, 20, 256)])
@pytest.mark.parametrize("reverse", [False, True])
def test_gru_direction_matches_torch(B, T, I, reverse):
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
