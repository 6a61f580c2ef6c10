#!/bin/bash
for lib in build/ab/edge_base.so build/ab/edge_vec.so; do
  LCTVQA_LIB=$PWD/$lib python tools/edge_debug.py 32 1 2 112 112 2>&1 | grep -v Warning | tail -24
done
python - <<'PY'
import torch
a=torch.load('/tmp/edge_dbg_edge_base.so.pt'); b=torch.load('/tmp/edge_dbg_edge_vec.so.pt')
for i,(x,y) in enumerate(zip(a,b)):
    print(i, tuple(x.shape), float((x-y).abs().max()), float(x.abs().max()))
PY
