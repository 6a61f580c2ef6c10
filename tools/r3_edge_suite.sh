#!/bin/bash
set -o pipefail
timeout -k 10 1100 python -m pytest tests/test_gpu_edge.py tests/test_gpu_darts.py tests/test_gpu_arch.py -q -m gpu -x --timeout 900 2>&1 | tail -15 > gpurun_out/r3_edge_suite.txt
rc=$?
cat gpurun_out/r3_edge_suite.txt
if [ $rc -ne 0 ]; then exit $rc; fi
python3 bench.py --workload darts_vqa --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_darts_now.json 2> gpurun_out/r3_darts_now.err || { tail -5 gpurun_out/r3_darts_now.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3_darts_now.json')); print('darts_vqa B=64:', d['ms_per_step'], 'ms/step', d['value'], d.get('roofline',{}).get('frac'), d.get('architect_step_ms'))"
