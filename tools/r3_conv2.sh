#!/bin/bash
# new conv-stage backward kernel: parity first, then timing against other builds of the library (same C signature)
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_edge.py -q -m gpu -x --timeout 600 2>&1 | tail -15 > gpurun_out/r3_conv2_test.txt
rc=$?
cat gpurun_out/r3_conv2_test.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/ab_edge.py lct_vqa_amd/_C/liblctvqa_hip.so $@ --batch 64 > gpurun_out/r3_conv2_ab.txt 2>&1 || { tail -5 gpurun_out/r3_conv2_ab.txt; exit 1; }
cut -c1-420 gpurun_out/r3_conv2_ab.txt
