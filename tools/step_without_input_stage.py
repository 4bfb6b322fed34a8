"""Diagnostic: bench.py's cnn_small step with the input stage (log-mel + SpecAugment) taken OUT of the loop -- features
are computed once up front, the timed steps are forward / loss / backward / clip / AdamW only.  The difference to bench.py's
ms_per_step is what the side-stream input stage still costs the critical path (CU / LDS / issue-slot sharing), and a
rocprofv3 --kernel-trace --stats run of this script gives every conv-stack kernel's duration WITHOUT that sharing.

    python tools/step_without_input_stage.py [--batch 512] [--dtype bf16] [--steps 50] [--warmup 10]
"""
import argparse
import contextlib
import json
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--dtype", choices=("bf16", "f32", "f16"), default="bf16")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    args = ap.parse_args()
    import torch
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer

    dev = "cuda:0"
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size = args.batch
    torch.manual_seed(1234)
    model = create_model("cnn_small", num_classes=2, pretrained=False, dropout=cfg.model.dropout,
                         act_dtype={"bf16": "bf16", "f16": "fp16"}.get(args.dtype, "fp32"))
    with contextlib.redirect_stdout(sys.stderr):
        trainer = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp(prefix="wwdiag_")), device=dev)
    trainer.model.train()
    pool = []
    for i in range(4):
        wave, y = make_synthetic_batch(args.batch, 24000, seed=1234 + i, device=dev)
        pool.append((trainer._features(wave, training=True, step=i), y))
    torch.cuda.synchronize()

    def step(i):
        feats, y = pool[i % len(pool)]
        for _ in trainer._step_native(feats, y, i):
            trainer.state.global_step += 1

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    list(trainer._flush_pending())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "cnn_small step, features precomputed (no log-mel in the loop)", "batch": args.batch,
                      "dtype": args.dtype, "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 4),
                      "samples_per_s": round(args.batch * args.steps / dt, 1)}))


if __name__ == "__main__":
    main()
