#!/bin/bash
for i in 1 2 3; do
  python bench.py --workload lct --dtype bf16 --steps 10 --no-cpu-baseline --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lct_bf16', d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
python bench.py --workload lct --dtype bf16 --steps 10 --no-cpu-baseline --warmup 3 --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lct_bf16 nopipe', d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
