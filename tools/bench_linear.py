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


def run16(M, K, N, dtype, out_dtype):
    """ww_gemm16_nt: operands already 16-bit in HBM (random normal operands -- not zeros: the clock depends on the data)."""
    a = (torch.randn(M, K, device=dev)).to(dtype)
    b = (torch.randn(N, K, device=dev) / K ** 0.5).to(dtype)
    t = timeit(lambda: nat.gemm16_nt(a, b, out_dtype=out_dtype))
    t_ref = timeit(lambda: torch.nn.functional.linear(a, b))
    fl = 2.0 * M * K * N
    return {"kernel": "ww_gemm16_nt", "M": M, "K": K, "N": N, "operands": str(dtype).split(".")[-1], "out": str(out_dtype).split(".")[-1],
            "us": round(t, 1), "TFLOPs": round(fl / t / 1e6, 1), "frac_of_2500_TF_dense_peak": round(fl / t / 1e6 / 2500, 3),
            "torch_linear_us(hipBLASLt, same operands)": round(t_ref, 1), "torch_TFLOPs": round(fl / t_ref / 1e6, 1)}


for M, K, N in ((16384, 1024, 1024), (8192, 8192, 8192), (4096, 4096, 4096), (19456, 64, 384), (2048, 576, 1024)):
    for dtype, od in ((torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float32)):
        print(json.dumps(run16(M, K, N, dtype, od)))
