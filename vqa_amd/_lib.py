"""ctypes loader for the C-ABI HIP library (vqa_amd/libvqa_hip.so; declarations in include/vqa_hip.h).

The library is loaded lazily at module scope, never stored on an nn.Module (the reference pickles whole
modules: vqa_model.py:296).  There is NO CPU fallback: if the library is missing, every op raises."""
import ctypes
import os
import re

import torch

_LIB = None
# VQA_HIP_LIB: another build of the same library (same-box A/B runs of kernel variants)
_PATH = os.environ.get('VQA_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvqa_hip.so')

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
        for name, (restype, argtypes) in declared_prototypes().items():
            fn = getattr(_LIB, name)
            fn.restype, fn.argtypes = restype, argtypes
    return _LIB


def dt(t):
    return _DT[t.dtype]


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _raw_stream():
    """The current HIP stream of the current device as an integer handle (the C call torch's own extensions use:
    ~0.3 us against ~9 us for torch.cuda.current_stream(); the step issues ~230 launches)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream():
    return ctypes.c_void_p(_raw_stream())


def check(status, what):
    if status != 0:
        raise VqaHipError(f'{what} failed with status {status}')


_FN = {}
_REC = None             # plan.Recorder of the pass being recorded (vqa_amd/plan.py), or None


def _conv(a):
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    if isinstance(a, (list, tuple)):            # a HOST array argument: n ints, or n device pointers (tensors / None)
        if a and all(isinstance(v, int) for v in a):
            return (ctypes.c_int * len(a))(*a)
        return (ctypes.c_void_p * len(a))(*[None if v is None else v.data_ptr() for v in a])
    return a


def call(name, *args):
    """Call an exported function: tensors -> device pointers, lists -> host arrays, ints/floats passed by value, stream
    prepended.  The ctypes prototypes are the declarations of include/vqa_hip.h (declared_prototypes(), set when the
    library is loaded), so every argument is converted to the declared C type whatever Python type a call site passes.
    While a plan is being recorded (vqa_amd/plan.py) every successful call is appended to it."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    raw = _raw_stream()
    status = fn(raw, *[_conv(a) for a in args])
    if status != 0:
        raise VqaHipError(f'{name} failed with status {status}', status)
    if _REC is not None:
        _REC.op(name, raw, args)


def fits(name, *args, **kw):
    """call() for the entry points that document status 1 as "this shape is not mine, use the general path" (the
    persistent recurrence): False on status 1 (nothing was launched), True on success, raises on anything else."""
    try:
        call(name, *args, **kw)
    except VqaHipError as e:
        if len(e.args) > 1 and e.args[1] == 1:
            return False
        raise
    return True


_HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'vqa_hip.h')
_PROTOS = None


def _ctype_of(param):
    """C parameter declaration -> ctypes type: any pointer is a void*, scalars by their declared type."""
    if '*' in param:
        return ctypes.c_void_p
    words = set(re.findall(r'[A-Za-z_]+', param)[:-1])          # the last identifier is the parameter's name
    for key, ct in (('double', ctypes.c_double), ('float', ctypes.c_float), ('long', ctypes.c_long),
                    ('unsigned', ctypes.c_uint), ('int', ctypes.c_int)):
        if key in words:
            return ct
    raise VqaHipError(f'include/vqa_hip.h: cannot map parameter {param!r} to a ctypes type')


def declared_prototypes():
    """{name: (restype, [argtypes])} of every function include/vqa_hip.h declares (comments stripped)."""
    global _PROTOS
    if _PROTOS is None:
        with open(_HEADER) as f:
            text = re.sub(r'/\*.*?\*/', ' ', f.read(), flags=re.S)
        protos = {}
        for name, params in re.findall(r'\bint\s+(vqa_\w+)\s*\(([^)]*)\)\s*;', text):
            plist = [p.strip() for p in params.split(',') if p.strip() and p.strip() != 'void']
            protos[name] = (ctypes.c_int, [_ctype_of(p) for p in plist])
        _PROTOS = protos
    return _PROTOS


def declared_symbols():
    """Names of every function include/vqa_hip.h declares."""
    return sorted(declared_prototypes())


# --- optional per-call device timing (bench.py's roofline leg): events on the launching stream around each call
_PROFILE = None


def profile_start():
    global _PROFILE
    _PROFILE = []


def profile_stop():
    """-> {entry point: [(ms, tag), ...]}"""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, tag, e0, e1 in rec:
        out.setdefault(name, []).append((e0.elapsed_time(e1), tag))
    return out


_call_raw = call


def call(name, *args, tag=None):  # noqa: F811
    """tag: a callable returning the launch's algorithmic work (kind, shape, flops, bytes) - only evaluated while
    profiling: the step issues ~150 launches and is host-bound, so nothing is formatted on the normal path."""
    if _PROFILE is None:
        return _call_raw(name, *args)
    tag = tag() if callable(tag) else tag
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _call_raw(name, *args)
    e1.record()
    _PROFILE.append((name, tag, e0, e1))
