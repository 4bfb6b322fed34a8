"""Isolated timing of ww_logmel_fwd for the FFT sizes other than 1024 (the general kernel k_logmel_any) beside the 1024 kernel.
usage: python tools/time_logmel_any.py [B]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from wakeword_trainer_home_amd import _native as nat
from wakeword_trainer_home_amd.data import make_synthetic_batch

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wave, _ = make_synthetic_batch(B, 24000, device=dev)
sa = nat.make_specaug_cfg(freq_mask_prob=0.5, time_mask_prob=0.5)
for n_fft, hop, n_mfcc in ((256, 160, 0), (512, 160, 0), (1024, 160, 0), (2048, 160, 0), (4096, 160, 0), (512, 160, 13), (1024, 160, 13)):
    cfg = nat.make_feat_cfg(n_fft=n_fft, hop=hop, n_mfcc=n_mfcc)
    for i in range(3):
        nat.logmel_fwd(wave, cfg, sa, seed=1, step=i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(20):
        nat.logmel_fwd(wave, cfg, sa, seed=1, step=i)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"n_fft {n_fft:4d} hop {hop} n_mfcc {n_mfcc:2d} B={B}: {us:8.1f} us per launch, {B / us:.2f} M clips/s")
