#!/bin/bash
for v in 1024 512 256 64; do
  LCTVQA_PW_GEMM_FROM=$v python3 tools/bench_darts.py --batch 64 --iters 4 2>&1 | tail -1 | sed "s/^/from=$v /"
done
for v in 1073741824 1024 256; do
  LCTVQA_PW_GEMM_FROM=$v python3 tools/bench_darts.py --batch 16 --iters 6 2>&1 | tail -1 | sed "s/^/from=$v /"
done
