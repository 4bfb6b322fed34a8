"""Times GRUWakeword (reference defaults: input 40, hidden 128, 2 layers, bidirectional) forward+backward on the native
kernels, next to torch.nn.GRU on the same GPU (MIOpen) for orientation."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat
from wakeword_trainer_home_amd.models import create_model

dev = "cuda:0"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def flops(B, T, I, H=128, L=2, nd=2):
    f = 0
    for k in range(L):
        isz = I if k == 0 else nd * H
        f += nd * 2.0 * B * T * (isz * 3 * H + H * 3 * H)
    return f       # forward; backward ~ 2x


for B, T in ((512, 151), (4096, 76), (4096, 151)):
    model = create_model("gru", input_size=40, dropout=0.3).to(dev).train()
    x = torch.randn(B, T, 40, device=dev)
    y = torch.randint(0, 2, (B,), device=dev)

    def step():
        model.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(model(x), y).backward()

    ms = timeit(step)
    nat.prof_enable(dev, ["gru"])
    step()
    torch.cuda.synchronize()
    prof = nat.prof_collect(dev)
    nat.prof_enable(dev, [])
    ref = torch.nn.GRU(40, 128, num_layers=2, batch_first=True, bidirectional=True, dropout=0.3).to(dev).train()
    fc = torch.nn.Linear(256, 2).to(dev)

    def step_ref():
        ref.zero_grad(set_to_none=True)
        _, hn = ref(x)
        torch.nn.functional.cross_entropy(fc(torch.cat([hn[-2], hn[-1]], 1)), y).backward()

    try:
        ms_ref = round(timeit(step_ref), 3)
    except RuntimeError as e:               # MIOpen gives up on the largest shape
        ms_ref = f"failed: {str(e)[:60]}"
    fl = 3 * flops(B, T, 40)
    print(json.dumps({"B": B, "T": T, "native_ms": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1),
                      "TFLOPs": round(fl / ms / 1e9, 2), "gru_class_ms": round(prof.get("gru", (0, 0))[0], 3),
                      "torch_nn_GRU_ms": ms_ref}))
