#!/bin/bash
# MobileNetV3 (BASELINE config 3, per-GPU batch 256) on the GPU box: layer tests, the graph-replayed step, its rocprofv3 kernel stats.
#   bash tools/prof_mnv3.sh LABEL   ->  gpurun_out/LABEL/{tests.log,step.json,kernel_stats.csv}
set -e
L=${1:-mnv3}
O=gpurun_out/$L
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_mobilenetv3.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 200 python tools/bench_models.py mobilenetv3 256 bf16 --graph > $O/step.json 2> $O/step.err
cat $O/step.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_models.py mobilenetv3 256 bf16 --graph > $O/step_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python tools/kernel_trace_by_grid.py $(ls $O/kt/*/*kernel_trace.csv | head -1) k_se k_dwg k_bn k_colstats k_gemm k_splitk k_colsum > $O/by_grid.txt
python tools/launches_per_replay.py $(ls $O/kt/*/*kernel_trace.csv | head -1) | tee $O/launches_per_replay.txt
rm -rf $O/kt
python tools/kernel_stats_top.py $O/kernel_stats.csv 40
