#!/bin/bash
# round-3 bench sweep on the GPU box: the headline and the own-metric workloads, one JSON line each under gpurun_out/
set -o pipefail
for spec in "bf16:--dtype bf16 --steps 20" "lct_bf16:--workload lct --dtype bf16 --steps 10" "lct_f32:--workload lct --steps 10" "f32:--steps 20"; do
  name=${spec%%:*}; args=${spec#*:}
  python bench.py $args --no-cpu-baseline --warmup 3 > gpurun_out/r3_b_$name.json 2> gpurun_out/r3_b_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r3_b_$name.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3_b_$name.json"))
    print("$name", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("conv_ms_per_step"))
except Exception as e:
    print("$name: no line", e)
PY
done
for spec in "resnet_f32:--workload resnet_gru --steps 8" "resnet_bf16:--workload resnet_gru --dtype bf16 --steps 10" "resnet_train:--workload resnet_gru --train-encoder --steps 4" "lct_train:--workload lct --train-encoder --steps 4" "darts_f32:--workload darts_vqa --batch 64 --steps 5" "darts_bf16:--workload darts_vqa --batch 64 --steps 5 --dtype bf16"; do
  name=${spec%%:*}; args=${spec#*:}
  python bench.py $args --no-cpu-baseline --warmup 2 > gpurun_out/r3_b_$name.json 2> gpurun_out/r3_b_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r3_b_$name.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3_b_$name.json"))
    print("$name", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"].get("architect_step_ms"))
except Exception as e:
    print("$name: no line", e)
PY
done
python tools/bench_darts.py --batch 16 --iters 6 --graph 2>&1 | tail -3
