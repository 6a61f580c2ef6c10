#!/bin/bash
for spec in "resnet_train_bf16:--workload resnet_gru --train-encoder --dtype bf16 --steps 4" "resnet_train_f32:--workload resnet_gru --train-encoder --steps 4" "lct_train_bf16:--workload lct --train-encoder --dtype bf16 --steps 4"; do
  name=${spec%%:*}; args=${spec#*:}
  python bench.py $args --no-cpu-baseline --warmup 2 > gpurun_out/r3_b_$name.json 2> gpurun_out/r3_b_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r3_b_$name.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3_b_$name.json"))
    print("$name", d["ms_per_step"], d["value"], d["loss"])
except Exception as e:
    print("$name: no line", e)
PY
done
