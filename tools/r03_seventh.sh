#!/bin/bash
set -e
O=gpurun_out/r03_j
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -q -m gpu -x -rP > $O/gpu_tests_rP.log 2>&1 || { grep -n "FAILED\|Error" $O/gpu_tests_rP.log | head; tail -40 $O/gpu_tests_rP.log | cut -c1-300; exit 1; }
tail -1 $O/gpu_tests_rP.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench_20_5.json'));print('20/5', d['value'], d['ms_per_step'], d['passes_ms_per_step'], d['roofline']['launch_us'], d['roofline']['side_stream']['alone']['launch_us'])"
