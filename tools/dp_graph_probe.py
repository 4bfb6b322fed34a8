"""One-rank RCCL process group + MobileNetV3 with the default config: the Trainer's unasked graph replay (training.hip_graph_auto)
captures the step WITH its in-stream gradient all-reduce; 6 steps, losses finite, graph captured.  usage: python tools/dp_graph_probe.py"""
import os
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
import torch
import torch.distributed as dist
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cfg = get_preset("cnn_small_logmel40")
cfg.training.epochs, cfg.training.batch_size, cfg.optimizer.warmup_epochs = 1, 16, 0
cfg.training.checkpoint_frequency = "best_only"
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(16 * 6, 1, 40, 151, generator=g)
y = (torch.rand(16 * 6, generator=g) < 0.3).long()
batches = [(x[16 * i:16 * i + 16], y[16 * i:16 * i + 16]) for i in range(6)]
t = Trainer(create_model("mobilenetv3", dropout=0.2), batches, batches[:1], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device="cuda:0")
losses = []
t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: losses.append(l)})())
t.train()
torch.cuda.synchronize()
print("backend", dist.get_backend(), "use_hip_graph", t.use_hip_graph, "captured", t._graph is not None, "losses", [round(v, 4) for v in losses])
assert t.use_hip_graph and t._graph is not None and all(v == v and abs(v) < 1e3 for v in losses)
dist.destroy_process_group()
print("ok")
