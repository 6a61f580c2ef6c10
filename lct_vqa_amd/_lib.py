"""ctypes binding of include/lctvqa.h (single source of truth: the header is parsed for the
argument types).  The product path has NO fallback: if the HIP library is missing or a call
fails, this raises."""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'lctvqa.h')
LIB_PATH = os.environ.get('LCTVQA_LIB') or os.path.join(HERE, '_C', 'liblctvqa_hip.so')      # LCTVQA_LIB: a variant build (A/B tools)

_CT = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float,
    'unsigned long long': ctypes.c_ulonglong, 'double': ctypes.c_double,
}


class HipLibraryMissing(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    out = {}
    for m in re.finditer(r'\b(int|long|const char\s*\*)\s+(lctvqa_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if 'char' in ret else _CT[ret]
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = ' '.join(a.split()[:-1])
                    argtypes.append(_CT[ty])
        out[name] = (restype, argtypes)
    return out


class _Lib:
    def __init__(self):
        self._dll = None
        self._sigs = parse_header()
        self._fns = {}
        self._wants_stream = {}
        self._memo = {}

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f'{LIB_PATH} not found: build it with `python -m lct_vqa_amd.build` '
                '(hipcc --offload-arch=gfx950).  lct_vqa_amd has no CPU/eager fallback.')
        dll = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in self._sigs.items():
            fn = getattr(dll, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        self._dll = dll
        return dll

    def last_error(self):
        return self.load().lctvqa_last_error().decode()

    def raw(self, name):
        fn = self._fns.get(name)
        if fn is None:
            fn = getattr(self.load(), 'lctvqa_' + name)
            self._fns[name] = fn
            self._wants_stream[name] = len(fn.argtypes) if fn.argtypes else 0
        return fn

    def query(self, name, *args):
        """Call a value-returning entry point (no error convention, no stream).  These are pure functions of their integer
        arguments (sizes of scratch buffers, "is this shape supported"), asked again for every launch: memoised."""
        key = (name, args)
        hit = self._memo.get(key)
        if hit is None:
            hit = self._memo[key] = self.raw(name)(*args)
        return hit

    def __call__(self, name, *args):
        """Call lctvqa_<name>; tensors are passed as device pointers; the current torch stream is
        appended when the C signature ends in `void* stream` and it was not given.  This sits on every launch of the
        launch-bound paths (LSTM steps, PC-DARTS primitives: ~10^4 calls per step), so it avoids per-call object
        construction: cached function objects, raw integer pointers, the raw current-stream handle."""
        fn = self._fns.get(name) or self.raw(name)
        conv = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
        if len(conv) == self._wants_stream[name] - 1:
            conv.append(_current_raw_stream())
        if _TRACE:                                                  # LCTVQA_TRACE_CALLS=<file>: fault hunting -- name + arguments of every call,
            _trace_call(name, args)                                 # the device drained after it: the last line names the kernel that faulted
        rc = fn(*conv)
        if _TRACE:
            torch.cuda.synchronize()
        if rc != 0 and fn.restype is ctypes.c_int:
            raise RuntimeError(f'lctvqa_{name} failed (rc={rc}): {self.last_error()}')
        return rc


_TRACE = os.environ.get('LCTVQA_TRACE_CALLS')


def _trace_call(name, args):
    desc = [f'{tuple(a.shape)}:{str(a.dtype)[6:]}@{a.data_ptr():x}+{a.numel() * a.element_size()}' if isinstance(a, torch.Tensor) else repr(a) for a in args]
    with open(_TRACE, 'a') as f:
        f.write(name + ' ' + ' '.join(desc) + '\n')


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream

    def _current_raw_stream():
        return _raw_stream(torch.cuda.current_device())
except AttributeError:                                          # pragma: no cover - older torch

    def _current_raw_stream():
        return torch.cuda.current_stream().cuda_stream


lib = _Lib()
