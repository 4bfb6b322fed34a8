#!/bin/bash
# bench.py variants of the headline step (run through gpurun): bash tools/regen_variants.sh LABEL
# -> gpurun_out/LABEL/bench_{f16,f32,graph,dist,augment}.json
set -e
L=${1:-variants}
O=gpurun_out/$L
mkdir -p $O
run() { python bench.py --no-cpu-baseline "${@:2}" > $O/bench_$1.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$1.json'));print('$1',d['value'],d['ms_per_step'])"; }
run bf16
run f16 --dtype f16
run f32 --dtype f32
run graph --graph
run dist --force-dist
run augment --augment
