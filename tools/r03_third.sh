#!/bin/bash
set -e
O=gpurun_out/r03_e
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_linear_mfma.py tests/test_mobilenetv3.py tests/test_hip_graph.py tests/test_gru.py tests/test_data_parallel_gpu.py tests/test_config_sizes.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -n "FAILED\|Error" $O/tests.log | head; tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for r in 512 256 128; do
  WW_DW_MIN_ROWS=$r timeout -k 10 200 python tools/bench_models.py mobilenetv3 256 bf16 --graph 2>> $O/model_steps.err | sed "s/^{/{\"dw_min_rows\": $r, /" >> $O/model_steps.jsonl
done
timeout -k 10 200 python tools/bench_models.py crnn 512 fp16 --graph >> $O/model_steps.jsonl 2>> $O/model_steps.err
timeout -k 10 200 python tools/bench_models.py crnn 4096 fp16 >> $O/model_steps.jsonl 2>> $O/model_steps.err
cat $O/model_steps.jsonl
