#!/bin/bash
# headline step with the input stage on N persistent log-mel workgroups (WW_LOGMEL_WGS overrides Trainer.input_stage_workgroups)
for w in "$@"; do
  WW_LOGMEL_WGS=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('wgs', $w, d['value'], d['ms_per_step'], d['passes_ms_per_step'], d['roofline']['side_stream']['in_step']['launch_us'])"
done
