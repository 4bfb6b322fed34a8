#!/bin/bash
set -e
O=gpurun_out/r03_f
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gru.py tests/test_hip_graph.py tests/test_config_sizes.py tests/test_fp16_mode.py -q -m gpu -x -rP > $O/tests.log 2>&1 || { grep -n "FAILED\|Error" $O/tests.log | head; tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for r in 16 8; do
  for cfg in "crnn 512 fp16" "crnn 512 fp16 --graph"; do
    WW_GRU_ROWS=$r timeout -k 10 200 python tools/bench_models.py $cfg 2>> $O/model_steps.err | sed "s/^{/{\"gru_rows\": $r, /" >> $O/model_steps.jsonl
  done
done
WW_GRU_ROWS=8 timeout -k 10 200 python tools/bench_models.py crnn 4096 fp16 2>> $O/model_steps.err | sed "s/^{/{\"gru_rows\": 8, /" >> $O/model_steps.jsonl
timeout -k 10 200 python tools/bench_models.py mobilenetv3 256 bf16 --graph >> $O/model_steps.jsonl 2>> $O/model_steps.err
cat $O/model_steps.jsonl
timeout -k 10 300 python tools/dp_overlap_probe.py > $O/dp_overlap_probe.json 2> $O/dp_overlap_probe.err || tail -5 $O/dp_overlap_probe.err
cat $O/dp_overlap_probe.json
