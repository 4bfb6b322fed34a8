"""Which ops of a model's training step issue device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer launches) or plain
elementwise adds: one eager step under torch.profiler with stacks.  usage: python tools/find_copies.py mobilenetv3 256 bf16"""
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
cfg.training.batch_size, cfg.training.hip_graph, cfg.training.hip_graph_auto = B, False, False
torch.manual_seed(0)
model = create_model(arch, dropout=0.3, **({"act_dtype": act} if arch == "crnn" else {"mode": act}))
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
tr.model.train()
pool = [make_synthetic_batch(B, 24000, seed=i, device=dev) for i in range(2)]
for i in range(3):
    tr._step_autograd_async(*pool[i % 2], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr._step_autograd_async(*pool[1], 3)
    torch.cuda.synchronize()
for avg in prof.key_averages(group_by_stack_n=8):
    if not any(t in avg.key for t in ("copy", "Memcpy", "clone", "aten::add", "aten::cat", "aten::contiguous")):
        continue
    if avg.device_time_total <= 0:
        continue
    stack = [s for s in avg.stack if "site-packages/torch" not in s][:4]
    print(f"{avg.count:4d} x {avg.key:28s} dev {avg.device_time_total:8.1f} us  <- " + " | ".join(x.strip() for x in stack))
