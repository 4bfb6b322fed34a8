"""Debug aid (GPU): run cnn_small layer by layer through the C-ABI and compare every
intermediate (y_l, g_l, dW_l, dgamma_l) with float64 autograd of the oracle."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import torch.nn.functional as F
from wakeword_trainer_home_amd import _native as nat
from oracle.cnn_small import CNNSmallOracle

DEV = "cuda:0"
B, Fd, T, p = [int(v) if i < 3 else float(v) for i, v in enumerate((sys.argv[1:] + ["4", "40", "151", "0.0"])[:4])]
torch.manual_seed(99)
model = CNNSmallOracle(dropout=p, dropout_seed=77).double()
with torch.no_grad():
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
gen = torch.Generator().manual_seed(5)
x = torch.randn(B, 1, Fd, T, generator=gen, dtype=torch.float64) * 2 - 4
dlog = torch.randn(B, 2, generator=gen, dtype=torch.float64) / B
model.train(); model.dropout_step = 4

# ---- oracle with retained intermediates
convs = [model.stem.conv] + [c for blk in model.blocks for c in (blk.dw, blk.pw)]
bns = [model.stem.bn] + [c for blk in model.blocks for c in (blk.dw_bn, blk.pw_bn)]
ys, zs = [], []
a = x
for conv, bn in zip(convs, bns):
    y = conv(a); y.retain_grad(); z = bn(y); z.retain_grad(); ys.append(y); zs.append(z); a = torch.relu(z)
pooled = a.mean(dim=(2, 3))
from oracle.cnn_small import dropout_keep_mask
if p > 0:
    keep = dropout_keep_mask(B, 64, p, 77, 4)
    pooled = pooled * torch.from_numpy(keep.astype(np.float64) / (1.0 - float(np.float32(p))))
out = model.classifier(pooled)
out.backward(dlog)

def cu(t): return t.detach().float().to(DEV).contiguous()
def nhwc(t): return t.detach().permute(0, 2, 3, 1).contiguous()
def rel(got, ref):
    got = got.detach().cpu().double().numpy(); ref = ref.detach().cpu().double().numpy()
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)

scratch = nat.layer_scratch(DEV)
keepalive = []
def mkbn(bn):
    t = [cu(bn.weight), cu(bn.bias), torch.zeros(64, device=DEV), torch.ones(64, device=DEV)]
    keepalive.append(t)
    return nat.make_bn(*t), t
Y, SS, MR = [], [], []
bnh, bt = mkbn(bns[0])
y, ss, mr = nat.conv_stem_fwd(cu(x), cu(convs[0].weight), bnh, scratch)
Y.append(y); SS.append(ss); MR.append(mr)
for l in range(1, 9):
    bnh, bt = mkbn(bns[l])
    fn = nat.dwconv3x3_fwd if l % 2 == 1 else nat.pwconv1x1_fwd
    y, ss, mr = fn(Y[-1], SS[-1], cu(convs[l].weight), bnh, scratch)
    Y.append(y); SS.append(ss); MR.append(mr)
for l in range(9):
    print(f"fwd y[{l}] rel err {rel(Y[l], nhwc(ys[l])):.2e}")
H, W = Y[0].shape[1], Y[0].shape[2]
pool = nat.gap_fwd(Y[8], SS[8], MR[8])
pd, logits = nat.head_fwd(pool, H * W, cu(model.classifier.weight), cu(model.classifier.bias), p, True, 77, 4)
print("logits", rel(logits, out))
dfc_w, dfc_b, dpool, coef8, dg8, db8 = nat.head_bwd(cu(dlog), pd, pool, H * W, cu(model.classifier.weight),
                                                    cu(bns[8].weight), MR[8], p, True, 77, 4)
print("dfc_w", rel(dfc_w, model.classifier.weight.grad), "dgamma8", rel(dg8, bns[8].weight.grad), "dbeta8",
      rel(db8, bns[8].bias.grad))
# reference coef for layer 8
def coef_ref(l):
    g = zs[l].grad; y = ys[l].detach(); gamma = bns[l].weight.detach()
    mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False); rstd = 1 / torch.sqrt(var + 1e-5)
    yhat = (y - mean[None, :, None, None]) * rstd[None, :, None, None]
    c1 = g.mean(dim=(0, 2, 3)); c2 = (g * yhat).mean(dim=(0, 2, 3)); A = gamma * rstd
    return torch.cat([A, -A * rstd * c2, A * (mean * rstd * c2 - c1)])
print("coef8 abs err", (coef8.cpu().double() - coef_ref(8)).abs().max().item(), "scale", coef_ref(8).abs().max().item())
print("dy8 check (ref dL/dy8) max", ys[8].grad.abs().max().item())
g = None
coef = coef8
for l in range(8, 0, -1):
    if l % 2 == 0:
        g_in, dw, coef_in, dg, db = nat.pwconv1x1_bwd(g, dpool if g is None else None, Y[l], SS[l] if g is None else None,
                                                      coef, Y[l - 1], SS[l - 1], MR[l - 1], cu(bns[l - 1].weight),
                                                      cu(convs[l].weight), scratch)
    else:
        g_in, dw, coef_in, dg, db = nat.dwconv3x3_bwd(g, Y[l], coef, Y[l - 1], SS[l - 1], MR[l - 1],
                                                      cu(bns[l - 1].weight), cu(convs[l].weight), scratch)
    print(f"bwd layer {l}: g_in(dL/dz[{l-1}]) {rel(g_in, nhwc(zs[l-1].grad)):.2e}  dW {rel(dw.reshape(-1), convs[l].weight.grad.reshape(-1)):.2e}"
          f"  dgamma[{l-1}] {rel(dg, bns[l-1].weight.grad):.2e} dbeta {rel(db, bns[l-1].bias.grad):.2e}"
          f"  coef_in abs {(coef_in.cpu().double() - coef_ref(l-1)).abs().max().item():.2e}/{coef_ref(l-1).abs().max().item():.2e}")
    # also: feed the EXACT g (from autograd) to see whether the error is inherited or local
    g, coef = g_in, coef_in
dw0 = nat.conv_stem_bwd(g, Y[0], coef, cu(x), scratch)
print("stem dW", rel(dw0.reshape(-1), convs[0].weight.grad.reshape(-1)))

# ---- detailed look at layer 8's g_in
g_in8, dw8, _, _, _ = nat.pwconv1x1_bwd(None, dpool, Y[8], SS[8], coef8, Y[7], SS[7], MR[7], cu(bns[7].weight),
                                         cu(convs[8].weight), scratch)
ref = nhwc(zs[7].grad).reshape(-1, 64).numpy()
got = g_in8.cpu().double().reshape(-1, 64).numpy()
err = np.abs(got - ref)
print("max ref", np.abs(ref).max(), "max err", err.max(), "mean err", err.mean(), "mean |ref|", np.abs(ref).mean())
print("err by channel (max) first 8:", err.max(0)[:8], " ... per-half:", err[:, :32].max(), err[:, 32:].max())
rows = np.arange(err.shape[0]) % 128
print("err by row%128 block of 32:", [float(err[(rows // 32) == i].max()) for i in range(4)])
print("frac elements with err>1e-3*max:", (err > 1e-3 * np.abs(ref).max()).mean())
bad = np.argwhere(err > 0.02 * np.abs(ref).max())
print("n bad", len(bad), "examples", bad[:10].tolist())
for (pp, cc) in bad[:6]:
    print(pp, cc, "got", got[pp, cc], "ref", ref[pp, cc], "z7", zs[7].detach().permute(0,2,3,1).reshape(-1,64)[pp, cc].item())
# using the same kernel with an explicit g (FROM_G path) built in torch
g8 = (dpool.cpu().double()[:, None, None, :] * (nhwc(zs[8]) > 0)).float().to(DEV).contiguous()
g_in8b, dw8b, _, _, _ = nat.pwconv1x1_bwd(g8, None, Y[8], None, coef8, Y[7], SS[7], MR[7], cu(bns[7].weight),
                                           cu(convs[8].weight), scratch)
print("FROM_G variant: g_in", rel(g_in8b, nhwc(zs[7].grad)), "vs FROM_POOL variant", rel(g_in8b, g_in8))
