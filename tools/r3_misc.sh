#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_edge.py -q -m gpu -x --timeout 600 -k "two_readers" 2>&1 | tail -5
timeout -k 10 300 python tools/r3_glue_trace.py > gpurun_out/r3_glue_trace.txt 2>&1; tail -62 gpurun_out/r3_glue_trace.txt | cut -c1-220
