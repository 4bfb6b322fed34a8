"""cProfile of the host side of one generic-path training step (model given by argv: mobilenetv3 | crnn | gru)."""
import cProfile
import contextlib
import pstats
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.data import make_synthetic_batch
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer

arch = sys.argv[1] if len(sys.argv) > 1 else "mobilenetv3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
cfg = get_preset("cnn_small_logmel40")
cfg.training.batch_size = B
model = create_model(arch, dropout=0.3)
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
tr.model.train()
step = tr._step_autograd_async if tr._async_autograd else tr._step_generic
pool = [make_synthetic_batch(B, 24000, seed=i, device=dev) for i in range(2)]
for i in range(3):
    step(*pool[i % 2], i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    step(*pool[i % 2], i)
torch.cuda.synchronize()
print(f"{arch} B={B}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(*pool[i % 2], i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
