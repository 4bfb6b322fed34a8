"""Isolated timing of the fused log-mel + SpecAugment launch (HIP events on the launch stream).
usage: python tools/time_logmel.py [B ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat
from wakeword_trainer_home_amd.data import make_synthetic_batch

dev = "cuda:0"
for B in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    wave, _ = make_synthetic_batch(B, 24000, device=dev)
    cfg = nat.make_feat_cfg()
    sa = nat.make_specaug_cfg(freq_mask_prob=0.5, time_mask_prob=0.5)
    for i in range(5):
        nat.logmel_fwd(wave, cfg, sa, seed=1, step=i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(50):
        nat.logmel_fwd(wave, cfg, sa, seed=1, step=i)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print(f"B={B}: {us:.1f} us per launch, {B * 151 / us:.1f} frames/us, {B / us * 1e6 / 1e6:.2f} M clips/s")
