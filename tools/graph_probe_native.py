"""Eager vs HIP-graph replay of the native cnn_small step on FEATURE inputs (no input-stage branch): what a linear chain of
graph nodes costs against host-issued launches when the host is not the bottleneck.  usage: python tools/graph_probe_native.py [B]"""
import json
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wakeword_trainer_home_amd.config import get_preset       # noqa: E402
from wakeword_trainer_home_amd.models import create_model     # noqa: E402
from wakeword_trainer_home_amd.training import Trainer        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda:0"
out = {"batch": B}
for mode in ("eager", "graph"):
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size = B
    cfg.training.hip_graph = mode == "graph"
    torch.manual_seed(0)
    model = create_model("cnn_small", dropout=0.3, act_dtype="bf16")
    t = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
    t.model.train()
    x = torch.randn(B, 1, 40, 151, device=dev) * 2 - 4
    y = (torch.rand(B, device=dev) < 0.3).long()

    def step(i, staged={}):
        prep = staged.pop(i, None) or t._prepare_native(x, y, i)
        staged[i + 1] = t._prepare_native(x, y, i + 1)
        t._step_native(None, None, i, prepared=prep)
    for i in range(12):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(12, 112):
        step(i)
    t._flush_pending()
    torch.cuda.synchronize()
    out[mode + "_ms"] = round((time.perf_counter() - t0) * 10, 4)
    out[mode + "_captured"] = t._graph is not None
print(json.dumps(out))
