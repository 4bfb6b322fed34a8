#!/bin/bash
# same-box A/B of bench.py: the built libwwhip.so against wakeword_trainer_home_amd/csrc/libwwhip_ab.so (another build), N alternating pairs
N=${1:-3}
P='import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["passes_ms_per_step"], d["roofline"]["side_stream"]["in_step"]["launch_us"])'
for i in $(seq $N); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" new
  WW_AB_LIB=wakeword_trainer_home_amd/csrc/libwwhip_ab.so python tools/ab_lib.py bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" ab
done
