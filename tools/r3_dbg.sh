#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_edge.py -q -m gpu -x --timeout 600 -k "darts_vqa_model_at_224" 2>&1 | grep -E "^E|assert|passed|failed" | head -30
LCTVQA_PW_GEMM_FROM=1073741824 timeout -k 10 600 python -m pytest tests/test_gpu_edge.py -q -m gpu -x --timeout 600 -k "darts_vqa_model_at_224" 2>&1 | tail -2
