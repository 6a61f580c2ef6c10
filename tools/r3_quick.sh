#!/bin/bash
# bf16-heads regression + the two bf16 bench lines
python -m pytest tests/test_gpu_bf16_heads.py tests/test_gpu_models.py "tests/test_gpu_full_size.py::test_ef_full_b4_bf16_against_reference_goldens" "tests/test_gpu_full_size.py::test_ef_full_b4_against_reference_goldens" -q -m gpu -x --timeout 600 > gpurun_out/r3_quick_tests.log 2>&1; tail -4 gpurun_out/r3_quick_tests.log
for spec in "bf16:--dtype bf16 --steps 20" "lct_bf16:--workload lct --dtype bf16 --steps 10" "lct_f32:--workload lct --steps 10"; do
  name=${spec%%:*}; args=${spec#*:}
  python bench.py $args --no-cpu-baseline --warmup 3 > gpurun_out/r3_q_$name.json 2> gpurun_out/r3_q_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r3_q_$name.err; }
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3_q_$name.json"))
    print("$name", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("conv_ms_per_step"))
except Exception as e:
    print("$name: no line", e)
PY
done
