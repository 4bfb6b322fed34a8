#!/bin/bash
# CRNN (BASELINE config 5, per-GPU batch 512, fp16) on the GPU box: the graph-replayed step and its rocprofv3 kernel stats.
#   bash tools/prof_crnn.sh LABEL  ->  gpurun_out/LABEL/{step.json,kernel_stats.csv}
set -e
L=${1:-crnn}
O=gpurun_out/$L
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python tools/bench_models.py crnn 512 fp16 --graph > $O/step.json 2> $O/step.err
cat $O/step.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_models.py crnn 512 fp16 --graph > $O/step_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python tools/kernel_trace_by_grid.py $(ls $O/kt/*/*kernel_trace.csv | head -1) k_gru k_gemm k_splitk k_colsum k_to16 k_dropout > $O/by_grid.txt
python tools/launches_per_replay.py $(ls $O/kt/*/*kernel_trace.csv | head -1) | tee $O/launches_per_replay.txt
rm -rf $O/kt
python tools/kernel_stats_top.py $O/kernel_stats.csv 40
