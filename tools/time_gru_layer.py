"""One GRU layer-direction (ww_gru_fwd / ww_gru_bwd) in isolation: microseconds per launch and per time step.
usage: python tools/time_gru_layer.py [B T I]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat

dev = "cuda:0"
B, T, I = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 76, 256)
H = 128
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, T, I, device=dev, generator=g)
w_ih = torch.randn(3 * H, I, device=dev, generator=g) * 0.05
w_hh = torch.randn(3 * H, H, device=dev, generator=g) * 0.05
b_ih, b_hh = torch.zeros(3 * H, device=dev), torch.zeros(3 * H, device=dev)
y = torch.empty(B, T, H, device=dev)
dy = torch.randn(B, T, H, device=dev, generator=g)
dx = torch.empty(B, T, I, device=dev)
ws = nat.gru_workspace(B, T, I, H, dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for mode in (torch.float32, torch.bfloat16):
    nat.prof_enable(dev, ["gru", "linear_mfma"])
    f = timed(lambda: nat.gru_fwd(x, w_ih, w_hh, b_ih, b_hh, y, ws, mode=mode))
    pf = nat.prof_collect(dev)
    bw = timed(lambda: nat.gru_bwd(x, w_ih, w_hh, dy, None, ws, dx=dx, mode=mode))
    pb = nat.prof_collect(dev)
    nat.prof_enable(dev, [])
    print(f"B={B} T={T} I={I} {str(mode)[6:]}: fwd {f:.0f} us ({f / T:.2f} us/step), bwd {bw:.0f} us ({bw / T:.2f} us/step); "
          f"classes fwd { {k: round(v[0] / 23 * 1e3) for k, v in pf.items()} } bwd { {k: round(v[0] / 23 * 1e3) for k, v in pb.items()} } (us per call)")
