#!/bin/bash
python -m pytest tests/test_gpu_edge.py tests/test_gpu_darts.py tests/test_gpu_darts_kernels.py tests/test_gpu_arch.py tests/test_gpu_bf16_heads.py -q -m gpu -x --timeout 900 > gpurun_out/r3_edge_tests.log 2>&1; tail -6 gpurun_out/r3_edge_tests.log
for dt in f32 bf16; do
  python bench.py --workload darts_vqa --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --dtype $dt 2> gpurun_out/r3_darts_$dt.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('darts_vqa', d['dtype'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['architect_step_ms'])" || tail -5 gpurun_out/r3_darts_$dt.err
done
