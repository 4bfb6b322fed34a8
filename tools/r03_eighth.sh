#!/bin/bash
set -e
O=gpurun_out/r03_k
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_linear_mfma.py tests/test_mobilenetv3.py -q -m gpu -x > $O/tests.log 2>&1 || { grep -n "FAILED\|Error" $O/tests.log | head; tail -40 $O/tests.log | cut -c1-300; exit 1; }
tail -1 $O/tests.log
for t in 0 1 2; do
  WW_GEMM_PAIR_TALL=$t timeout -k 10 200 python tools/bench_models.py mobilenetv3 256 bf16 --graph 2>> $O/model_steps.err | sed "s/^{/{\"pair_tall\": $t, /" >> $O/model_steps.jsonl
done
cat $O/model_steps.jsonl | cut -c1-240
