"""One-rank RCCL process group + MobileNetV3 at its per-GPU batch 256 (BASELINE config 3): what the two forms of the gradient
all-reduce cost the step on ONE rank -- i.e. their fixed overheads (stream hand-offs, hooks); the payoff of the overlap (a 6 MB
all-reduce hidden under the early layers' backward) exists only at N > 1 and is NOT measurable on this box.
    eager + one in-stream all-reduce after the backward        (training.dp_overlap = False)
    eager + tail of the bucket reduced from a post-accumulate hook under the early layers' backward (dp_overlap = True)
    graph replay (in-stream all-reduce captured with the step; Trainer default for this model under RCCL)
usage: python tools/dp_overlap_probe.py  -> one JSON line"""
import contextlib
import gc
import json
import os
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
import torch
import torch.distributed as dist
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.data import make_synthetic_batch
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer

dev = "cuda:0"
torch.cuda.set_device(0)
saved = os.dup(1)
os.dup2(2, 1)                                  # RCCL's banner goes to stderr
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
warm = torch.zeros(1, device=dev)
dist.all_reduce(warm)
torch.cuda.synchronize()
os.dup2(saved, 1)
B = 256
pool = [make_synthetic_batch(B, 24000, seed=i, device=dev) for i in range(2)]
out = {"model": "mobilenetv3", "batch": B, "ranks": 1, "backend": dist.get_backend()}
for name, graph, overlap in (("eager_in_stream", False, False), ("eager_overlapped", False, True), ("graph_in_stream", True, False)):
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size = B
    cfg.training.hip_graph, cfg.training.hip_graph_auto, cfg.training.dp_overlap = graph, False, overlap
    torch.manual_seed(0)
    model = create_model("mobilenetv3", dropout=0.3, mode="bf16")
    with contextlib.redirect_stdout(sys.stderr):
        tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
    tr.model.train()
    for i in range(6):
        tr._step_autograd_async(*pool[i % 2], i)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    n = 30
    for i in range(n):
        tr._step_autograd_async(*pool[i % 2], 6 + i)
    torch.cuda.synchronize()
    gc.enable()
    out[name] = {"ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 3), "collective": tr.last_collective,
                 "graph": tr._graph is not None, "bucket_MB": round(model.flat_grad.numel() * 4 / 2 ** 20, 2)}
    del tr, model
    torch.cuda.empty_cache()
print(json.dumps(out))
dist.destroy_process_group()
