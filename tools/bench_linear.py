"""Times the MFMA dense head (Linear(576,1024)+Hardswish+Dropout+Linear(1024,2), BASELINE config 3's per-GPU batch 256 and
the global batch 2048) and a large GEMM, forward and backward, with HIP events on the launch stream."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat

dev = "cuda:0"


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def run(M, K, N, mode):
    md = torch.float32 if mode == "fp32" else torch.bfloat16
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    y, pre = nat.linear_mfma_fwd(x, w, b, act=nat.LIN_HARDSWISH, dropout_p=0.3, mode=md, want_pre=True)
    t_f = timeit(lambda: nat.linear_mfma_fwd(x, w, b, act=nat.LIN_HARDSWISH, dropout_p=0.3, mode=md, want_pre=True))
    t_b = timeit(lambda: nat.linear_mfma_bwd(x, w, pre, dy, act=nat.LIN_HARDSWISH, dropout_p=0.3, mode=md))
    t_ref_f = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * K * N
    return {"M": M, "K": K, "N": N, "mode": mode, "fwd_us": round(t_f, 1), "bwd_us": round(t_b, 1),
            "fwd_TFLOPs": round(fl / t_f / 1e6, 2), "bwd_TFLOPs": round(2 * fl / t_b / 1e6, 2),
            "torch_linear_fwd_us(hipBLASLt fp32)": round(t_ref_f, 1)}


for M, K, N in ((256, 576, 1024), (2048, 576, 1024), (16384, 1024, 1024)):
    for mode in ("fp32", "bf16"):
        print(json.dumps(run(M, K, N, mode)))
