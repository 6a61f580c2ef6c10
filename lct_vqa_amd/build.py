"""Build the gfx950 shared library of the VQA hot path in-tree with hipcc.

    python -m lct_vqa_amd.build         # -> lct_vqa_amd/_C/liblctvqa_hip.so

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, '_C')
LIB = os.path.join(OUT_DIR, 'liblctvqa_hip.so')
SOURCES = ['api.hip', 'conv3x3.hip', 'conv3x3_bf16.hip', 'conv3x3_bwd.hip', 'conv3x3_wino.hip', 'conv3x3_wino4.hip', 'conv3x3_wino_wgrad.hip', 'darts.hip', 'gemm.hip', 'gemm_bf16.hip', 'mixed_edge.hip', 'mixed_edge_bf16.hip', 'ops.hip', 'resnet.hip', 'train_bf16.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', 
         '-Wno-unused-result', '-Wno-unused-value', '-Wno-inline-asm', '-I' + CSRC]


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stamp():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    stamp_file = os.path.join(OUT_DIR, 'stamp.txt')
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(OUT_DIR, s.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
