#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_darts_kernels.py tests/test_gpu_darts.py -q -m gpu -x --timeout 600 2>&1 | tail -6 || exit 1
bash tools/r3_darts_nums.sh
