import sys, time, tempfile, contextlib
from pathlib import Path
sys.path.insert(0, '.')
import torch
from wakeword_trainer_home_amd import _native as nat
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.data import make_synthetic_batch
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer
act = sys.argv[1]
cfg = get_preset("cnn_small_logmel40"); cfg.training.batch_size = 512
torch.manual_seed(0)
model = create_model("crnn", dropout=0.3, act_dtype=act)
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device="cuda:0")
tr.model.train()
pool = [make_synthetic_batch(512, 24000, seed=i, device="cuda:0") for i in range(2)]
for i in range(5): tr._step_autograd_async(*pool[i % 2], i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30): tr._step_autograd_async(*pool[i % 2], 5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(act, "host issue ms/step", round((t1 - t0) / 30 * 1e3, 3), "wall ms/step", round((t2 - t0) / 30 * 1e3, 3))
