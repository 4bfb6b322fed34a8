"""Probe: capture forward + native loss + backward of a HIP-backed autograd model in a HIP graph (torch.cuda.CUDAGraph)
and compare replay time with the eager step.  usage: python tools/graph_probe.py [arch] [B]"""
import contextlib
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
wave, y = make_synthetic_batch(B, 24000, seed=0, device=dev)
feats = tr._features(wave, training=True, step=0)
s_in, s_tg = feats.clone(), y.to(dev).clone()


def fwd_bwd():
    tr.optimizer.zero_grad(set_to_none=True)
    loss = tr.criterion(tr.model(s_in), s_tg)
    loss.backward()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fwd_bwd()
torch.cuda.synchronize()
print(f"eager fwd+bwd: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = fwd_bwd()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
ref = [p.grad.clone() for p in tr.model.parameters()]
l0 = float(loss)
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay fwd+bwd: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms, loss {l0:.6f}")
fwd_bwd_loss = float(fwd_bwd())
torch.cuda.synchronize()
err = max(float((a - p.grad).abs().max()) for a, p in zip(ref, tr.model.parameters()))
print(f"eager loss {fwd_bwd_loss:.6f}; max |grad_graph - grad_eager| = {err:.3e}")
