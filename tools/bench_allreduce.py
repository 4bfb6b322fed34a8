"""What one small all-reduce costs inside a stream of dependent kernels on ONE rank (RCCL through torch.distributed):
the exposed collective of the data-parallel step.  usage: python tools/bench_allreduce.py"""
import json
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
buf = torch.zeros(20547, device=dev)
a = torch.randn(4096, 4096, device=dev)
dist.all_reduce(buf)
torch.cuda.synchronize()


def work():                       # ~100 us of dependent main-stream work
    return (a @ a[:, :512]).sum()


def run(kind, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        work()
        if kind == "sync":
            dist.all_reduce(buf, op=dist.ReduceOp.AVG)
        elif kind == "async":
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True).wait()
        elif kind == "two_async":
            w1 = dist.all_reduce(buf[10000:], op=dist.ReduceOp.AVG, async_op=True)
            work()
            w2 = dist.all_reduce(buf[:10000], op=dist.ReduceOp.AVG, async_op=True)
            w1.wait()
            w2.wait()
        elif kind == "two_none":
            work()
        buf.mul_(1.0)             # the consumer (clip/optimizer) on the main stream
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = {}
for kind in ("none", "sync", "async", "two_none", "two_async"):
    run(kind, 20)
    out[kind] = round(run(kind), 2)
out["exposed_sync_us"] = round(out["sync"] - out["none"], 2)
out["exposed_async_us"] = round(out["async"] - out["none"], 2)
out["exposed_two_async_us"] = round(out["two_async"] - out["two_none"], 2)
print(json.dumps(out))
dist.destroy_process_group()
