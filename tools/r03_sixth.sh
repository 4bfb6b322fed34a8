#!/bin/bash
set -e
O=gpurun_out/r03_i
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gru.py tests/test_linear_mfma.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -n "FAILED\|Error" $O/tests.log | head; tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for sp in 32 64 128; do
  WW_GRU_SPLITS=$sp timeout -k 10 200 python tools/bench_models.py crnn 512 fp16 --graph 2>> $O/model_steps.err | sed "s/^{/{\"gru_splits\": $sp, /" >> $O/model_steps.jsonl
done
cat $O/model_steps.jsonl | cut -c1-260
