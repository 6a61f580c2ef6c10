#!/bin/bash
set -o pipefail
timeout -k 10 1100 python -m pytest tests/test_gpu_edge.py tests/test_gpu_darts.py tests/test_gpu_arch.py -q -m gpu -x --timeout 900 2>&1 | tail -5 || exit 1
python3 bench.py --workload darts_vqa --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('darts_vqa B=64:', d['ms_per_step'], 'architect', d['config']['architect_step_ms'])"
