"""Capture + replay the sync-free training step of one model as a HIP graph and compare with eager steps (debug aid for
tests/test_hip_graph.py).  usage: python tools/graph_capture_probe.py ARCH [overlap|serial] [B]"""
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wakeword_trainer_home_amd.config import get_preset       # noqa: E402
from wakeword_trainer_home_amd.models import create_model     # noqa: E402
from wakeword_trainer_home_amd.training import Trainer        # noqa: E402

arch = sys.argv[1]
overlap = (sys.argv[2] if len(sys.argv) > 2 else "overlap") == "overlap"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = "cuda:0"
res = {}
for mode in ("eager", "graph"):
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size, cfg.training.hip_graph = B, mode == "graph"
    cfg.optimizer.mixed_precision = False
    torch.manual_seed(0)
    model = create_model(arch, dropout=0.3)
    for m in model.modules():
        if hasattr(m, "overlap_directions"):
            m.overlap_directions = overlap
    t = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
    t.model.train()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 1, 40, 151, device=dev, generator=g) * 2 - 4
    y = (torch.rand(B, device=dev, generator=g) < 0.3).long()
    step = t._step_native if t.native else t._step_autograd_async
    losses = []
    for i in range(6):
        print(mode, "step", i, "graph" if t._graph is not None else "eager", flush=True)
        for d in step(x, y, i):
            losses.append(d[1])
    losses += [d[1] for d in t._flush_pending()]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6, 26):
        step(x, y, i)
    t._flush_pending()
    torch.cuda.synchronize()
    res[mode] = (losses, (time.perf_counter() - t0) / 20 * 1e3, {k: v.clone() for k, v in model.state_dict().items()})
same = res["eager"][0] == res["graph"][0] and all(torch.equal(a, b) for a, b in zip(res["eager"][2].values(), res["graph"][2].values()))
print(f"{arch} overlap={overlap} B={B}: bit-identical={same} eager {res['eager'][1]:.3f} ms graph {res['graph'][1]:.3f} ms")
print(res["eager"][0], res["graph"][0])
