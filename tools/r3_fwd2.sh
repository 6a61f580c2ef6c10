#!/bin/bash
set -o pipefail
timeout -k 10 300 python tools/ab_edge_exact.py build/ab/edge_fwdv1.so lct_vqa_amd/_C/liblctvqa_hip.so 2>&1 | tail -12
timeout -k 10 300 python tools/ab_edge_exact.py build/ab/edge_fwdv1.so build/ab/edge_fwd2s1.so 2>&1 | tail -3
timeout -k 10 300 python tools/ab_edge.py build/ab/edge_fwdv1.so lct_vqa_amd/_C/liblctvqa_hip.so build/ab/edge_fwd2s1.so --batch 64 > gpurun_out/r3_fwd2_ab.txt 2>&1 || { tail -5 gpurun_out/r3_fwd2_ab.txt; exit 1; }
cut -c1-330 gpurun_out/r3_fwd2_ab.txt
