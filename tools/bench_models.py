"""BASELINE configs 3 and 5 on one GPU (MobileNetV3 with the reference head at the per-GPU and the global batch; CRNN (conv front-end + 2-layer bidirectional GRU) training step -- log-mel front end,
forward, native loss, backward, clip, AdamW -- at batch 4096 (and 512), inputs resident in HBM.  Secondary measurement:
bench.py's headline line stays BASELINE config 2."""
import contextlib
import gc
import json
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat
from wakeword_trainer_home_amd.config import get_preset
from wakeword_trainer_home_amd.data import make_synthetic_batch
from wakeword_trainer_home_amd.models import create_model
from wakeword_trainer_home_amd.training import Trainer

dev = "cuda:0"
CONFIGS = (("crnn", 512, "bf16"), ("crnn", 4096, "bf16"), ("gru", 4096, "fp32"), ("mobilenetv3", 256, "bf16"),
           ("mobilenetv3", 2048, "bf16"))
GRAPH = "--graph" in sys.argv                 # replay the step as a captured HIP graph (Trainer hip_graph mode)
argv = [a for a in sys.argv[1:] if a != "--graph"]
if argv:                                      # e.g.  bench_models.py crnn 4096 fp16 [--graph]
    CONFIGS = ((argv[0], int(argv[1]), argv[2]),)
for arch, B, act in CONFIGS:
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size = B
    cfg.training.hip_graph, cfg.training.hip_graph_auto = GRAPH, False       # the eager line is eager for every model
    torch.manual_seed(0)
    kw = {"act_dtype": act} if arch == "crnn" else {"mode": act}
    model = create_model(arch, dropout=0.3, **kw)
    with contextlib.redirect_stdout(sys.stderr):
        tr = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp()), device=dev)
    tr.model.train()
    pool = [make_synthetic_batch(B, 24000, seed=i, device=dev) for i in range(2)]
    classes = ["logmel_specaug", "conv_stem_fwd", "dwconv3x3_fwd", "pwconv1x1_fwd", "pwconv1x1_bwd", "dwconv3x3_bwd", "conv_stem_bwd",
               "finalize", "gru", "linear_mfma", "nhwc_layers"]

    def step(i):
        (tr._step_autograd_async if tr._async_autograd else tr._step_generic)(*pool[i % 2], i)

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    n = 30 if B <= 1024 else 12
    gc.collect()
    gc.disable()          # as timeit does: a generation-2 collection inside the loop is a one-off 50-80 ms host pause (seen: one 76 ms
    t0 = time.perf_counter()      # step among thirty 2.9 ms ones made "crnn 512 fp16" read 5.3 ms instead of 2.94)
    for i in range(n):
        step(5 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gc.enable()
    nat.prof_enable(dev, classes)
    tr.use_hip_graph = False                   # the per-class event timing needs host-issued launches
    step(100)
    torch.cuda.synchronize()
    prof = {k: round(v[0], 3) for k, v in nat.prof_collect(dev).items()}
    nat.prof_enable(dev, [])
    print(json.dumps({"model": arch, "batch": B, "conv_storage": act if arch == "crnn" else None, "matrix_mode": act,
                      "hip_graph": tr._graph is not None, "loss_scale": tr.scaler.get_scale() if tr.loss_scale is not None else None,
                      "ms_per_step": round(dt * 1e3, 3),
                      "samples_per_s": round(B / dt, 1), "class_ms_one_step": prof}))
    del tr, model, pool
    torch.cuda.empty_cache()
