#!/bin/bash
timeout -k 10 300 python tools/ab_edge.py build/ab/edge_base.so build/ab/edge_vec.so > gpurun_out/r3_ab_edge2.txt 2>&1; tail -7 gpurun_out/r3_ab_edge2.txt
python -m pytest tests/test_gpu_edge.py tests/test_gpu_darts.py tests/test_gpu_darts_kernels.py -q -m gpu -x --timeout 900 > gpurun_out/r3_edge_tests2.log 2>&1; tail -4 gpurun_out/r3_edge_tests2.log
