#!/bin/bash
timeout -k 10 400 python tools/ab_edge.py lct_vqa_amd/_C/liblctvqa_hip.so $@ --batch 64 > gpurun_out/r3_ab_only.txt 2>&1 || { tail -5 gpurun_out/r3_ab_only.txt; exit 1; }
cut -c1-520 gpurun_out/r3_ab_only.txt
