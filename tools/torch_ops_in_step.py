"""Which torch-level kernels (copies, adds, fills ...) a model's training step still launches beside the library's own, and from
where: torch.profiler over two eager steps, aten ops with a device kernel, grouped by python call site.
usage: python tools/torch_ops_in_step.py mobilenetv3 256 bf16 | crnn 512 fp16"""
import contextlib
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.profiler import ProfilerActivity, profile
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.data import make_synthetic_batch
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer

arch, B, act = sys.argv[1], int(sys.argv[2]), sys.argv[3]
dev = "cuda:0"
cfg = get_preset("cnn_small_logmel40")
cfg.training.batch_size = B
cfg.training.hip_graph, cfg.training.hip_graph_auto = False, False
torch.manual_seed(0)
model = create_model(arch, dropout=0.3, **({"act_dtype": act} if arch == "crnn" else {"mode": act}))
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
tr.model.train()
pool = [make_synthetic_batch(B, 24000, seed=i, device=dev) for i in range(2)]
step = lambda i: (tr._step_autograd_async if tr._async_autograd else tr._step_generic)(*pool[i % 2], i)
for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(4)
    step(5)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    site = next((f for f in ev.stack if "wakeword_trainer_home_amd" in f or "/tools/" in f), ev.stack[0] if ev.stack else "?")
    k = (ev.name, site.split("wakeword_trainer_home_amd/")[-1][:90])
    r = rows.setdefault(k, [0, 0.0])
    r[0] += 1
    r[1] += sum(kk.duration for kk in ev.kernels)
print(f"{arch} B={B} {act}: aten ops with device kernels over 2 steps")
for (name, site), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {n:3d} x {name:28s} {us:8.1f} us   {site}")
