#!/bin/bash
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv_first" 2>&1 | tail -3
cat > /tmp/t_first.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from lct_vqa_amd._lib import lib
lib.load()
B, H, W = 256, 224, 224
x = torch.randn(B, 3, H, W, device='cuda'); w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2; b = torch.randn(64, device='cuda')
y = torch.empty(B, H, W, 64, device='cuda')
for _ in range(3): lib('conv3x3_first_nchw', x, w, b, y, B, H, W, 64, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): lib('conv3x3_first_nchw', x, w, b, y, B, H, W, 64, 1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('LCTVQA_FIRST_MFMA=%s: %.3f ms  %.2f TB/s written' % (os.environ.get('LCTVQA_FIRST_MFMA', '1'), ms, y.numel() * 4 / ms / 1e9), float(y.sum()))
PY
LCTVQA_FIRST_MFMA=0 python /tmp/t_first.py
LCTVQA_FIRST_MFMA=1 python /tmp/t_first.py
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32', d['ms_per_step'], d['value'], d['roofline']['conv_ms_per_step'])"
LCTVQA_FIRST_MFMA=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 (VALU conv1_1)', d['ms_per_step'], d['value'], d['roofline']['conv_ms_per_step'])"
