#!/bin/bash
for lib in build/ab/edge_oldpw.so build/ab/edge_olddd.so; do
  LCTVQA_LIB=$PWD/$lib python tools/edge_debug.py 32 1 2 112 112 2>&1 | grep -E "^lib|dx |_ops.4.op.1" 
done
