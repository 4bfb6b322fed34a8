"""Diagnostic: which resource does a co-resident workgroup take from each conv-stack kernel?  A persistent dummy kernel with the
log-mel kernel's per-CU footprint (tools/corun/corun.hip: 256 threads, ~100 VGPRs, 41 KB LDS, one workgroup per CU) runs on a
side stream in one of five modes (0 idle / 1 LDS traffic / 2 VALU / 3 vector loads / 4 LDS+VALU) while the kernel under test
is timed on the main stream.  Build the co-runner first:  hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/corun/corun.hip -o
tools/corun/libcorun.so"""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat

dev = "cuda:0"
lib = C.CDLL(str(Path(__file__).resolve().parent / "corun" / "libcorun.so"))
lib.corun_launch.argtypes = [C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
B = 512
act = torch.bfloat16
y_in = torch.randn(B, 20, 76, 64, device=dev).to(act)
g = torch.randn(B, 20, 76, 64, device=dev).to(act)
ss = torch.cat([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3])
mr = torch.cat([torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.7])
t = [torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2, torch.zeros(64, device=dev), torch.ones(64, device=dev)]
bn = nat.make_bn(*t)
scratch = nat.layer_scratch(dev)
wdw = torch.randn(64, 1, 3, 3, device=dev) * 0.3
wpw = torch.randn(64, 64, 1, 1, device=dev) * 0.2
coef = torch.randn(192, device=dev) * 0.1
KERNELS = {
    "pw_bwd": lambda: nat.pwconv1x1_bwd(g, None, y_in, None, coef, y_in, ss, mr, t[0], wpw, scratch),
    "pw_fwd": lambda: nat.pwconv1x1_fwd(y_in, ss, wpw, bn, scratch),
    "dw_bwd": lambda: nat.dwconv3x3_bwd(g, y_in, coef, y_in, ss, mr, t[0], wdw, scratch),
    "dw_fwd": lambda: nat.dwconv3x3_fwd(y_in, ss, wdw, bn, scratch),
}
src = torch.randn(1 << 16, device=dev)
out = torch.zeros(256, device=dev)
side = torch.cuda.Stream(device=dev)
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count


def corun(mode, iters, grid, where=None):
    rc = lib.corun_launch(mode, iters, grid, src.data_ptr(), out.data_ptr(), side.cuda_stream, where.data_ptr() if where is not None else None)
    assert rc == 0, rc


def time_fn(fn, n=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


# calibrate the co-runner: iterations for ~6 ms per mode
iters = {}
for mode in range(5):
    corun(mode, 200, n_cu)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(side)
    corun(mode, 2000, n_cu)
    b.record(side)
    b.synchronize()
    us_per_iter = a.elapsed_time(b) * 1e3 / 2000
    iters[mode] = int(6000 / us_per_iter)
    print(json.dumps({"corunner_mode": mode, "us_per_iteration": round(us_per_iter, 3)}))
for name, fn in KERNELS.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    row = {"kernel": name, "alone_us": round(time_fn(fn), 1)}
    for mode, label in enumerate(("idle_footprint", "lds", "valu", "vmem", "lds_valu")):
        for grid, gl in ((n_cu, "1_per_cu"),):
            t_side0, t_side1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_side0.record(side)
            corun(mode, iters[mode], grid)
            t_side1.record(side)
            torch.cuda._sleep(400000)                 # ~0.2 ms on the main stream: the co-runner (idle side stream) is resident by now
            row[f"{label}"] = round(time_fn(fn), 1)
            torch.cuda.synchronize()
            row[f"{label}_corunner_ms"] = round(t_side0.elapsed_time(t_side1), 2)     # must exceed the timed span (8 launches)
    print(json.dumps(row))


# ---- where does the dispatcher put 256 persistent workgroups?  (a) on an idle GPU, (b) launched while a conv kernel is running
from collections import Counter
for label, busy in (("idle GPU", False), ("beside k_pw_bwd", True), ("beside k_dw_bwd", True)):
    where = torch.zeros(n_cu, dtype=torch.int32, device=dev)
    fn = KERNELS["dw_bwd"] if "dw" in label else KERNELS["pw_bwd"]
    if busy:
        for _ in range(6):
            fn()
    corun(0, 50, n_cu, where)
    if busy:
        for _ in range(6):
            fn()
    torch.cuda.synchronize()
    per_cu = Counter(where.cpu().tolist())
    hist = Counter(per_cu.values())
    print(json.dumps({"placement": label, "workgroups": n_cu, "distinct_cus": len(per_cu), "cus_by_workgroups_hosted": dict(sorted(hist.items()))}))
