"""Phase timing of k_logmel from s_memtime stamps (one wave of one workgroup, its third item).  Needs a library built with
-DWW_LOGMEL_STAMPS as wakeword_trainer_home_amd/csrc/libwwhip_ab.so:
    cd csrc && hipcc <CXXFLAGS> -DWW_LOGMEL_STAMPS -c ww_frontend.hip -o /tmp/f.o && hipcc -shared <other objects> /tmp/f.o -o libwwhip_ab.so
usage: python tools/logmel_stamps.py [workgroups: 0 = full device | 256 ...] [waves 4|8]"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 2:
    os.environ["WW_LOGMEL_WAVES"] = sys.argv[2]
from wakeword_trainer_home_amd import _native as nat
nat._LIB_PATH = ROOT / "wakeword_trainer_home_amd" / "csrc" / "libwwhip_ab.so"
import torch
from wakeword_trainer_home_amd.data import make_synthetic_batch

dev = "cuda:0"
wgs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wave, _ = make_synthetic_batch(512, 24000, device=dev)
cfg = nat.make_feat_cfg()
sa = nat.make_specaug_cfg(freq_mask_prob=0.5, time_mask_prob=0.5)
nat.set_logmel_workgroups(dev, wgs)
for i in range(3):
    nat.logmel_fwd(wave, cfg, sa, seed=1, step=i)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib = nat.load()
lib.ww_debug_logmel_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.ww_debug_logmel_stamps(buf) == 0
t = list(buf)
names = {1: "span staged (+barrier)", 2: "r0 window loads", 3: "r0 fft16 #1", 4: "r0 twiddles #1 (table loads)", 5: "r0 exchange 1",
         6: "r0 fft16 #2 + twiddles", 7: "r0 exchange 2", 8: "r0 pass 3 + power", 9: "r0 mel bands", 10: "r1 window loads",
         11: "r1 fft16 #1", 12: "r1 twiddles #1", 13: "r1 exchange 1", 14: "r1 fft16 #2 + twiddles", 15: "r1 exchange 2",
         16: "r1 pass 3 + power", 17: "r1 mel bands", 18: "barrier after the rounds", 19: "log pass (+barrier)",
         20: "masked write-out", 21: "closing barrier"}
prev = t[0]
print(f"workgroups {wgs or 'full device'}; item total {t[21] - t[0]} ticks of s_memtime")
for i in range(1, 22):
    print(f"  {names[i]:32s} {t[i] - prev:8d}")
    prev = t[i]
