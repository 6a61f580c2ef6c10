#!/bin/bash
for a in "" "--trunk-split 2" "--trunk-split 2 --head-after 0" "--head-after 10" "--head-after 6"; do
  python bench.py $a --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 [$a]', d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
