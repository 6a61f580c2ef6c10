#!/bin/bash
python3 bench.py --workload darts_vqa --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_darts_now.json 2> gpurun_out/r3_darts_now.err || { tail -5 gpurun_out/r3_darts_now.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3_darts_now.json')); print('darts_vqa B=64:', d['ms_per_step'], 'ms/step', d['value'], d.get('roofline',{}).get('frac'), d['config'])"
python3 tools/bench_darts.py --batch 16 --iters 6 2>&1 | tail -2
python3 tools/bench_darts.py --batch 64 --iters 4 2>&1 | tail -2
