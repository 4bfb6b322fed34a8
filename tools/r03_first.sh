#!/bin/bash
# round-3 first GPU call: driver-flag bench line, 50/10 line, spawner test, model baselines
set -e
O=gpurun_out/r03_a
mkdir -p $O
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python -c "import json;d=json.load(open('$O/bench_20_5.json'));print('20/5', d['value'], d['ms_per_step'], d['passes_ms_per_step'], d['roofline']['launch_us'])"
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_50_10.json 2> $O/bench_50_10.err
python -c "import json;d=json.load(open('$O/bench_50_10.json'));print('50/10', d['value'], d['ms_per_step'], d['passes_ms_per_step'])"
timeout -k 10 800 python -m pytest tests/test_data_parallel_gpu.py -x -q -m gpu -k "spawner or plumbing" > $O/spawner_test.log 2>&1 || { tail -40 $O/spawner_test.log; exit 1; }
tail -2 $O/spawner_test.log
for cfg in "mobilenetv3 256 bf16 --graph" "crnn 512 fp16" "crnn 512 fp16 --graph"; do
  timeout -k 10 200 python tools/bench_models.py $cfg >> $O/model_steps.jsonl 2>> $O/model_steps.err
done
cat $O/model_steps.jsonl
