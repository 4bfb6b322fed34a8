#!/bin/bash
# same-box comparison of where the look-ahead input stage may start (bench.py WW_INPUT_GATE): none | fwd | bwd | mid, N rounds
N=${1:-3}
P='import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["passes_ms_per_step"], d["roofline"]["side_stream"]["in_step"]["launch_us"])'
for i in $(seq $N); do
  for g in none fwd bwd mid; do
    if [ $g = none ]; then python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" $g
    else WW_INPUT_GATE=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" $g; fi
  done
done
