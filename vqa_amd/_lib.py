"""ctypes loader for the C-ABI HIP library (vqa_amd/libvqa_hip.so; declarations in include/vqa_hip.h).

The library is loaded lazily at module scope, never stored on an nn.Module (the reference pickles whole
modules: vqa_model.py:296).  There is NO CPU fallback: if the library is missing, every op raises."""
import ctypes
import os

import torch

_LIB = None
_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvqa_hip.so')

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


class VqaHipError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_PATH):
            raise VqaHipError(f'{_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950). '
                              'The VQA hot path has no CPU fallback.')
        _LIB = ctypes.CDLL(_PATH)
    return _LIB


def dt(t):
    return _DT[t.dtype]


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status, what):
    if status != 0:
        raise VqaHipError(f'{what} failed with status {status}')


def call(name, *args):
    """Call an exported function: tensors -> device pointers, ints/floats passed by value, stream prepended."""
    fn = getattr(lib(), name)
    conv = [stream()]
    for a in args:
        if a is None or isinstance(a, torch.Tensor):
            conv.append(ptr(a))
        elif isinstance(a, float):
            conv.append(ctypes.c_float(a))
        else:
            conv.append(ctypes.c_long(int(a)))   # x86-64: int and long params both read it correctly
    check(fn(*conv), name)
