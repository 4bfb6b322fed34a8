#!/bin/bash
# full GPU suite with the measured values printed (-rP), then the CRNN / MobileNetV3 steps eager vs graph
set -e
O=gpurun_out/r03_d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu -rP -x > $O/gpu_tests_rP.log 2>&1 || { grep -n "FAILED\|Error" $O/gpu_tests_rP.log | head; tail -30 $O/gpu_tests_rP.log; exit 1; }
tail -1 $O/gpu_tests_rP.log
for cfg in "crnn 512 fp16" "crnn 512 fp16 --graph" "crnn 512 bf16" "crnn 4096 fp16" "gru 4096 fp32" "mobilenetv3 256 bf16 --graph"; do
  timeout -k 10 200 python tools/bench_models.py $cfg >> $O/model_steps.jsonl 2>> $O/model_steps.err
done
cat $O/model_steps.jsonl
