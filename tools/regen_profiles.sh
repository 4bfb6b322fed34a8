#!/bin/bash
# Regenerate the judged artefacts of bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/regen_profiles.sh LABEL [notests]
# -> gpurun_out/LABEL/{gpu_tests.log,bench.json,kernel_stats.csv,pmc_traffic.json,no_input_stage.json,...}
# Copy bench.json / kernel_stats.csv / pmc_traffic.json into profiles/ afterwards (profiles/README.md names them).
set -e
L=${1:-run}
O=gpurun_out/$L
mkdir -p $O
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  # -rP: the parity tests print their measured values (the bounds are <= 10x those); kept as parity_measured.txt
  python -m pytest tests -q -m gpu -rP > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
  tail -1 $O/gpu_tests.log
  grep -h "^config \|storage: max\|max |loss_HIP" $O/gpu_tests.log > $O/parity_measured.txt || true
fi
python tools/step_without_input_stage.py > $O/no_input_stage.json 2>/dev/null
cat $O/no_input_stage.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python tools/kernel_stats_top.py $O/kernel_stats.csv 14
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $L
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/kt $O/pmc_fetch $O/pmc_write
# last, so that roofline.traffic comes from the PMC file written above (same kernel-source digest); the driver's command line
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print('bench',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['launch_us'])"
