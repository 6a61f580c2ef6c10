"""lct_vqa_amd -- MI355X-native (gfx950) implementation of the LCT-VQA train-step hot path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/lctvqa.h)
  _lib.py      ctypes binding (fails loudly when the library is missing; no fallback)
  ops.py       torch.autograd.Function wrappers (PyTorch = device memory/streams/graph plumbing)
  vgg.py       VGG19 module shell (torchvision key names) running the HIP trunk
  basic_vqa/   mirror of the reference's plugin surface: config, models, models_lct,
               model_factory, architect_factory (same names, signatures, state_dict keys)
  parallel.py  one-process-per-GPU data parallel step (RCCL all-reduce of flat gradient buckets)
"""
__version__ = '0.1.0'
