#!/bin/bash
for lib in build/ab/edge_vec.so build/ab/edge_nopwb.so; do
  echo "== $lib"
  LCTVQA_LIB=$PWD/$lib python -m pytest tests/test_gpu_edge.py -q -m gpu --timeout 900 -k "encoder_shapes or ragged" 2>&1 | grep -E "^E  |passed|failed" | head -12
done
