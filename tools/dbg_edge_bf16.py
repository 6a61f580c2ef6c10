"""Debug: the bf16-storage edge case [16-1-1-224-224] as the first and as the second call of a process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_edge as T
from lct_vqa_amd._lib import lib
lib.load()
args = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else [16, 1, 1, 224, 224]
for rep in range(3):
    try:
        T._case(lib, *args, True, True, 516 + rep, pre_bf16=True)
        print('rep', rep, 'ok', flush=True)
    except AssertionError as e:
        print('rep', rep, 'FAIL', str(e)[:1500], flush=True)
