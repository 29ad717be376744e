"""Record once, replay with ONE C-ABI call (csrc/plan.hip, vqa_run_plan).

The host side of a training step was ~3.3 ms of Python / ctypes for ~4.1 ms of GPU work: ~230 binding calls at ~15 us
each around launches that cost ~4 us.  A `Recorder` rides along while the ordinary per-launch path of an autograd
Function runs once: every `_lib.call` is appended to a POD word list, every temporary the pass allocates (`empty`)
gets an offset in one scratch arena, and every pointer argument is stored relocatable, as (base index, byte offset) -
base 0 is the arena, the others are the tensors the caller names (`bind`: inputs, parameters, the gradient buffer, the
forward pass's arena).  Later passes with the same key allocate one arena, fill the table of base addresses and make one
call; the launches, their order, their arguments and their streams are those of the recorded pass, so the results are
the same bits (tests/test_gpu_plan.py).

What a key must cover: everything the recorded path branches on - shapes, dtypes, modes, the environment knobs - is
the caller's business (vqa_amd/cnn.py builds it).  A plan is never used while `_lib` profiles per-call timings.

The library side stays stateless: plans live here, in Python, as ctypes arrays; nothing native is stored on a module.
"""
import ctypes
import os

import torch

from . import _lib as L

# VQA_PLAN=0: always the per-launch path
ENABLED = os.environ.get('VQA_PLAN', '1') != '0'
MAGIC = 0x56514150314c
K_INT, K_F32, K_F64, K_PTR, K_NULL, K_PTRARR, K_INTARR = range(7)
OP_SYNC = -1
ALIGN = 256

_plans = {}
_active = None          # the Recorder of the pass being recorded (one at a time: a pass records synchronously)


def lookup(key):
    return _plans.get(key) if ENABLED and L._PROFILE is None else None


def recording():
    return _active


def clear():
    _plans.clear()


class PlanError(L.VqaHipError):
    pass


class Recorder:
    """with Recorder(key, device) as rec: ... the ordinary path, using empty() / stream_wait() / _lib.call ..."""

    def __init__(self, key, device):
        self.key, self.device = key, device
        self.words, self.n_ops = [], 0
        self.size = 0
        self.tables = []            # per base: list of (ptr, nbytes, byte offset inside the base)
        self.names = ['arena']      # base names, index = base index
        self.tables.append([])
        self.keep = []              # recorded temporaries stay alive until the end: no address is handed out twice
        self.streams = {L._raw_stream(): 0}
        self.stream_names = ['main']
        self.exports = {}
        self.live = []              # (ptr, nbytes, arena offset) of the exported tensors: what outlives the recording
        self.ok = True

    # ---- bases
    def alloc(self, shape, dtype):
        t = torch.empty(shape, device=self.device, dtype=dtype)
        nbytes = t.numel() * t.element_size()
        self.tables[0].append((t.data_ptr(), nbytes, self.size))
        self.size += -(-max(nbytes, 1) // ALIGN) * ALIGN
        self.keep.append(t)
        return t

    def bind(self, name, tensor):
        """An external base: a tensor (its storage range from data_ptr) the replay gets a fresh address for."""
        self.names.append(name)
        self.tables.append([(tensor.data_ptr(), tensor.numel() * tensor.element_size(), 0)])
        self.keep.append(tensor)

    def bind_table(self, name, table):
        """An external base made of several recorded ranges: the allocations of another pass's arena."""
        self.names.append(name)
        self.tables.append(list(table))

    def stream(self, name, s):
        self.streams[s.cuda_stream] = len(self.stream_names)
        self.stream_names.append(name)

    def export(self, name, t):
        """Remember where a tensor of this pass lives in the arena, so that a replay can hand out a view of it."""
        b, off = self._ref(t.data_ptr())
        if b != 0:
            raise PlanError(f'export {name}: not an arena tensor')
        self.exports[name] = (off, tuple(t.shape), t.dtype)
        self.live.append((t.data_ptr(), t.numel() * t.element_size(), off))

    def _ref(self, ptr):
        # the LATEST matching range wins inside a table (none is ever reused while recording, see keep)
        for b in range(len(self.tables)):
            for p0, n, off in reversed(self.tables[b]):
                if p0 <= ptr < p0 + max(n, 1):
                    return b, off + (ptr - p0)
        raise PlanError(f'plan {self.key[0]}: a pointer argument belongs to no base (allocate it with plan.empty or '
                        f'bind it)')

    def _pref(self, t):
        if t is None:
            return -1
        b, off = self._ref(t.data_ptr())
        return (b << 48) | off

    # ---- ops
    def op(self, name, raw_stream, args):
        sidx = self.streams.get(raw_stream)
        if sidx is None:
            raise PlanError(f'plan {self.key[0]}: {name} was issued on a stream the recorder does not know')
        entry = L.lib().vqa_plan_opcode(name.encode())
        if entry < 0:
            raise PlanError(f'{name} has no plan thunk (scripts/gen_plan_thunks.py)')
        argtypes = L.declared_prototypes()[name][1][1:]
        w = [entry, sidx, len(args)]
        for a, ct in zip(args, argtypes):
            if isinstance(a, torch.Tensor):
                w += [K_PTR, self._pref(a)]
            elif a is None:
                w += [K_NULL, 0]
            elif isinstance(a, (list, tuple)):
                if all(isinstance(v, int) and not isinstance(v, bool) for v in a) and a:
                    w += [K_INTARR, len(a)] + [int(v) for v in a]
                else:
                    w += [K_PTRARR, len(a)] + [self._pref(v) for v in a]
            elif ct is ctypes.c_float:
                w += [K_F32, int.from_bytes(bytes(ctypes.c_float(a)), 'little')]
            elif ct is ctypes.c_double:
                w += [K_F64, int.from_bytes(bytes(ctypes.c_double(a)), 'little', signed=True)]
            else:
                w += [K_INT, int(a)]
        self.words += w
        self.n_ops += 1

    def sync(self, waiter_raw, signal_raw):
        self.words += [OP_SYNC, self.streams[waiter_raw], 1, K_INT, self.streams[signal_raw]]
        self.n_ops += 1

    # ---- context
    def __enter__(self):
        global _active
        if _active is not None:
            raise PlanError('nested plan recording')
        _active = self
        L._REC = self
        return self

    def __exit__(self, et, ev, tb):
        global _active
        _active = None
        L._REC = None
        if et is None and self.ok:
            _plans[self.key] = Plan(self)
        self.keep = []
        return False


class Plan:
    def __init__(self, rec):
        words = [MAGIC, rec.n_ops] + rec.words
        self.n_words = len(words)
        self.words = (ctypes.c_long * self.n_words)(*words)
        self.arena_bytes = max(rec.size, ALIGN)
        self.names = list(rec.names)
        self.stream_names = list(rec.stream_names)
        self.exports = dict(rec.exports)
        # the recorded pass's own addresses of what it exported: a later pass (its backward) records against them.  Only
        # the exported tensors - the other temporaries are freed when the recording ends, and their addresses may be
        # handed out again (to the backward's incoming gradient, say)
        self.table = list(rec.live)
        self.n_ops = rec.n_ops
        self._bases = (ctypes.c_void_p * len(self.names))()
        self._streams = (ctypes.c_void_p * len(self.stream_names))()
        self._fn = L.lib().vqa_run_plan

    def arena(self, device):
        return torch.empty(self.arena_bytes, device=device, dtype=torch.uint8)

    def view(self, arena, name):
        off, shape, dtype = self.exports[name]
        n = 1
        for s in shape:
            n *= s
        return arena[off:off + n * dtype.itemsize].view(dtype).view(shape)

    def run(self, arena, bases, side_streams=()):
        """bases: data pointers (ints) of the external bases in the order they were bound; side_streams: raw handles
        of the named side streams in the order they were declared.  The current stream is stream 0."""
        b = self._bases
        b[0] = arena.data_ptr()
        if len(bases) != len(self.names) - 1 or len(side_streams) != len(self.stream_names) - 1:
            raise PlanError('plan called with the wrong number of bases / streams')
        for i, p in enumerate(bases):
            b[i + 1] = p
        s = self._streams
        s[0] = L._raw_stream()
        for i, h in enumerate(side_streams):
            s[i + 1] = h
        status = self._fn(self.words, self.n_words, b, len(self.names), s, len(self.stream_names))
        if status != 0:
            raise L.VqaHipError(f'vqa_run_plan failed: status {status & 255} at op {status >> 8} of {self.n_ops}')


def empty(shape, dtype, device):
    """torch.empty, or - while a pass is being recorded - an allocation of the plan's arena."""
    if _active is not None:
        return _active.alloc(shape, dtype)
    return torch.empty(shape, device=device, dtype=dtype)


def stream_wait(waiter, signaler):
    """`waiter` (a torch stream) waits for everything issued so far on `signaler`; recorded as one plan op."""
    waiter.wait_event(signaler.record_event())
    if _active is not None:
        _active.sync(waiter.cuda_stream, signaler.cuda_stream)
