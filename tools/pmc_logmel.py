"""PMC helper: run one kernel family in isolation so rocprofv3 --pmc rows are easy to read.
usage: python tools/pmc_logmel.py [logmel|dw_fwd|dw_bwd|pw_fwd|pw_bwd] [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat
from wakeword_trainer_home_amd.data import make_synthetic_batch

what = sys.argv[1] if len(sys.argv) > 1 else "logmel"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
B = 512
if what == "logmel":
    wave, _ = make_synthetic_batch(B, 24000, device=dev)
    cfg = nat.make_feat_cfg()
    sa = nat.make_specaug_cfg(freq_mask_prob=0.5, time_mask_prob=0.5)
    for i in range(iters):
        nat.logmel_fwd(wave, cfg, sa, seed=1, step=i)
else:
    act = torch.bfloat16
    y_in = torch.randn(B, 20, 76, 64, device=dev).to(act)
    ss = torch.cat([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3])
    mr = torch.cat([torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.7])
    t = [torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2, torch.zeros(64, device=dev), torch.ones(64, device=dev)]
    bn = nat.make_bn(*t)
    scratch = nat.layer_scratch(dev)
    wdw = torch.randn(64, 1, 3, 3, device=dev) * 0.3
    wpw = torch.randn(64, 64, 1, 1, device=dev) * 0.2
    coef = torch.randn(192, device=dev) * 0.1
    g = torch.randn(B, 20, 76, 64, device=dev).to(act)
    for i in range(iters):
        if what == "dw_fwd":
            nat.dwconv3x3_fwd(y_in, ss, wdw, bn, scratch)
        elif what == "pw_fwd":
            nat.pwconv1x1_fwd(y_in, ss, wpw, bn, scratch)
        elif what == "dw_bwd":
            nat.dwconv3x3_bwd(g, y_in, coef, y_in, ss, mr, t[0], wdw, scratch)
        elif what == "pw_bwd":
            nat.pwconv1x1_bwd(g, None, y_in, None, coef, y_in, ss, mr, t[0], wpw, scratch)
        elif what == "gap":
            nat.gap_fwd(y_in, ss, mr)
        elif what == "all":
            nat.gap_fwd(y_in, ss, mr)
            nat.dwconv3x3_fwd(y_in, ss, wdw, bn, scratch)
            nat.pwconv1x1_fwd(y_in, ss, wpw, bn, scratch)
            nat.dwconv3x3_bwd(g, y_in, coef, y_in, ss, mr, t[0], wdw, scratch)
            nat.pwconv1x1_bwd(g, None, y_in, None, coef, y_in, ss, mr, t[0], wpw, scratch)
torch.cuda.synchronize()
print("done", what)
