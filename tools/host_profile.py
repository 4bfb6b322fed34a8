"""Host-side cost of one native step (cProfile over a host-bound run: small batch, so the GPU never throttles the host)."""
import cProfile
import contextlib
import pstats
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.data import make_synthetic_batch
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
cfg = get_preset("cnn_small_logmel40")
cfg.training.batch_size = B
model = create_model("cnn_small", act_dtype="bf16")
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
tr.model.train()
pool = [make_synthetic_batch(B, 24000, seed=i, device=dev) for i in range(2)]
staged = {}


def step(i):
    if i not in staged:
        staged[i] = tr._prepare_native(*pool[i % 2], i)
    prep = staged.pop(i)
    staged[i + 1] = tr._prepare_native(*pool[(i + 1) % 2], i + 1)
    for _ in tr._step_native(None, None, i, prepared=prep):
        tr.state.global_step += 1


for i in range(20):
    step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(20, 220):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"B={B}: host issue {1e3 * (t1 - t0) / 200:.3f} ms/step, with drain {1e3 * (time.perf_counter() - t0) / 200:.3f}")
pr = cProfile.Profile()
pr.enable()
for i in range(220, 420):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
