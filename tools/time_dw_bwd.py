"""Time k_dw_bwd / k_dw_fwd alone at several batch sizes (how much the uneven last round of the persistent grid costs).
usage: python tools/time_dw_bwd.py B [B ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat

dev = "cuda:0"
act = torch.bfloat16
for B in [int(a) for a in sys.argv[1:]] or [323, 512, 646]:
    y_in = torch.randn(B, 20, 76, 64, device=dev).to(act)
    g = torch.randn(B, 20, 76, 64, device=dev).to(act)
    ss = torch.cat([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3])
    mr = torch.cat([torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.7])
    t = [torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2, torch.zeros(64, device=dev), torch.ones(64, device=dev)]
    bn = nat.make_bn(*t)
    scratch = nat.layer_scratch(dev)
    wdw = torch.randn(64, 1, 3, 3, device=dev) * 0.3
    coef = torch.randn(192, device=dev) * 0.1
    res = {}
    for name, fn in (("dw_bwd", lambda: nat.dwconv3x3_bwd(g, y_in, coef, y_in, ss, mr, t[0], wdw, scratch)),
                     ("dw_fwd", lambda: nat.dwconv3x3_fwd(y_in, ss, wdw, bn, scratch))):
        for _ in range(3):
            fn()
        cls = "dwconv3x3_bwd" if name == "dw_bwd" else "dwconv3x3_fwd"
        nat.prof_enable(dev, [cls])
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        ms, n = nat.prof_collect(dev)[cls]
        nat.prof_enable(dev, [])
        res[name] = ms / n * 1e3
    items = B * 19
    print(f"B={B}: items {items} = {items / 6144:.2f} rounds of 6144 slots (3 WG/CU) / {items / 8192:.2f} of 8192 (4 WG/CU); "
          f"dw_bwd {res['dw_bwd']:.1f} us ({res['dw_bwd'] / B * 1e3:.1f} ns/clip), dw_fwd {res['dw_fwd']:.1f} us ({res['dw_fwd'] / B * 1e3:.1f} ns/clip)")
