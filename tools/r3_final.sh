#!/bin/bash
# final pass of the round: whole GPU suite, the darts numbers, then the profile summaries of the same tree
set -o pipefail
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --timeout 900 2>&1 | tail -6 > gpurun_out/r3_full_suite.txt
rc=$?
cat gpurun_out/r3_full_suite.txt
if [ $rc -ne 0 ]; then exit $rc; fi
python3 bench.py --workload darts_vqa --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_darts_now.json 2> gpurun_out/r3_darts_now.err || { tail -5 gpurun_out/r3_darts_now.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3_darts_now.json')); print('darts_vqa B=64:', d['ms_per_step'], 'ms/step', d['value'], d['roofline']['frac'], 'architect', d['config']['architect_step_ms'])"
python3 tools/bench_darts.py --batch 16 --iters 6 2>&1 | tail -1
bash tools/profile_round.sh r03 > gpurun_out/r3_profile_round.log 2>&1 || { tail -5 gpurun_out/r3_profile_round.log; exit 1; }
tail -3 gpurun_out/r3_profile_round.log
