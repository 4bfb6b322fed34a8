set -e
mkdir -p gpurun_out/r02n
python -m pytest tests/test_hip_kernels.py tests/test_full_size.py tests/test_trainer_gpu.py tests/test_hip_graph.py tests/test_config_sizes.py -q -m gpu -x 2>&1 | tail -4
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r02n/bench_new_$i.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/r02n/bench_new_$i.json'));s=d['roofline']['side_stream'];print(d['value'],d['ms_per_step'],'alone',s['alone']['launch_us'],'in_step',s['in_step']['launch_us'], s['in_step']['where'][-60:])"
done
python bench.py --no-cpu-baseline --graph > gpurun_out/r02n/bench_new_graph.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/r02n/bench_new_graph.json'));print('graph',d['value'],d['ms_per_step'])"
python bench.py --no-cpu-baseline --dtype f16 > gpurun_out/r02n/bench_new_f16.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/r02n/bench_new_f16.json'));print('f16',d['value'],d['ms_per_step'])"
python bench.py --no-cpu-baseline --dtype f32 > gpurun_out/r02n/bench_new_f32.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/r02n/bench_new_f32.json'));print('f32',d['value'],d['ms_per_step'])"
python tools/bench_models.py --model crnn --batch 512 2>/dev/null | tail -2
python tools/bench_models.py --model mobilenetv3 --batch 256 2>/dev/null | tail -2
