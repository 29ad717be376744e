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
# VQA_PLAN_EVENTS=0: an ordering point creates and destroys its event inside vqa_run_plan instead of reusing one of ours
OWN_EVENTS = os.environ.get('VQA_PLAN_EVENTS', '1') != '0'
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
    """with Recorder(key, device) as rec: ... the ordinary path, using empty() / stream_wait() / _lib.call ...

    Arena layout.  Offsets are assigned when the recording ends, from the temporaries' LIFETIMES in the recorded op
    sequence: a temporary's bytes are handed out again once its last op is behind - as torch's caching allocator
    recycles the blocks of the per-launch path - instead of every temporary getting bytes of its own.  That is not
    about the footprint (288 GB): a dead tensor whose lines still sit in the 256 MB Infinity Cache is overwritten
    there and never travels to HBM, while a layout without reuse writes every dead byte back (same-box A/B of the whole
    step: 4.29 ms with one offset per temporary against 4.245 per-launch).  A temporary that a side stream touches
    stays reserved until the main stream has waited for that stream (the ordering point after its last use); an
    exported one (saved for the backward) is never recycled."""

    def __init__(self, key, device):
        self.key, self.device = key, device
        self.ops = []               # ('call', entry, stream idx, [encoded args]) | ('sync', waiter idx, signal idx)
        self.allocs = []            # [ptr, nbytes, first op, last op, {stream: its first op}, exported]
        self.tables = [None]        # per external base: list of (ptr, nbytes, byte offset inside the base); [0] = arena
        self.names = ['arena']      # base names, index = base index
        self.keep = []              # recorded temporaries stay alive until the end: no address is handed out twice
        self.streams = {L._raw_stream(): 0}
        self.stream_names = ['main']
        self.exports = {}           # name -> (alloc id, delta, shape, dtype)
        self.patches = {}           # name -> [(entry, argument index)]
        self.ok = True

    # ---- bases
    def alloc(self, shape, dtype):
        t = torch.empty(shape, device=self.device, dtype=dtype)
        self.allocs.append([t.data_ptr(), t.numel() * t.element_size(), len(self.ops), len(self.ops), {}, False])
        self.keep.append(t)
        return t

    def bind(self, name, tensor):
        """An external base: a tensor (its storage range from data_ptr) the replay gets a fresh address for."""
        self.names.append(name)
        self.tables.append([(tensor.data_ptr(), tensor.numel() * tensor.element_size(), 0)])
        self.keep.append(tensor)

    def bind_table(self, name, table):
        """An external base made of several recorded ranges: the exported tensors of another pass's arena."""
        self.names.append(name)
        self.tables.append(list(table))

    def stream(self, name, s):
        self.streams[s.cuda_stream] = len(self.stream_names)
        self.stream_names.append(name)

    def export(self, name, t):
        """Remember where a tensor of this pass lives in the arena, so that a replay can hand out a view of it."""
        ref = self._ref(t.data_ptr(), None)
        if ref[0] != 'a':
            raise PlanError(f'export {name}: not an arena tensor')
        self.allocs[ref[1]][5] = True
        self.exports[name] = (ref[1], ref[2], tuple(t.shape), t.dtype, t.data_ptr(), t.numel() * t.element_size())

    def _ref(self, ptr, sidx):
        """-> ('a', alloc id, delta) for an arena temporary (its use by stream `sidx` at the current op is noted), or
        ('b', base index, byte offset) for an external base."""
        for i in range(len(self.allocs) - 1, -1, -1):
            a = self.allocs[i]
            if a[0] <= ptr < a[0] + max(a[1], 1):
                if sidx is not None:
                    a[3] = len(self.ops)
                    a[4].setdefault(sidx, len(self.ops))
                return ('a', i, ptr - a[0])
        for b in range(1, len(self.tables)):
            for p0, n, off in reversed(self.tables[b]):
                if p0 <= ptr < p0 + max(n, 1):
                    return ('b', b, off + (ptr - p0))
        raise PlanError(f'plan {self.key[0]}: a pointer argument belongs to no base (allocate it with plan.empty or '
                        f'bind it)')

    # ---- ops
    def op(self, name, raw_stream, args):
        sidx = self.streams.get(raw_stream)
        if sidx is None:
            raise PlanError(f'plan {self.key[0]}: {name} was issued on a stream the recorder does not know')
        entry = L.lib().vqa_plan_opcode(name.encode())
        if entry < 0:
            raise PlanError(f'{name} has no plan thunk (scripts/gen_plan_thunks.py)')
        argtypes = L.declared_prototypes()[name][1][1:]
        pref = lambda t: None if t is None else self._ref(t.data_ptr(), sidx)
        enc = []
        for a, ct in zip(args, argtypes):
            if isinstance(a, torch.Tensor):
                enc.append((K_PTR, pref(a)))
            elif a is None:
                enc.append((K_NULL, 0))
            elif isinstance(a, (list, tuple)):
                if all(isinstance(v, int) and not isinstance(v, bool) for v in a) and a:
                    enc.append((K_INTARR, [int(v) for v in a]))
                else:
                    enc.append((K_PTRARR, [pref(v) for v in a]))
            elif ct is ctypes.c_float:
                enc.append((K_F32, int.from_bytes(bytes(ctypes.c_float(a)), 'little')))
            elif ct is ctypes.c_double:
                enc.append((K_F64, int.from_bytes(bytes(ctypes.c_double(a)), 'little', signed=True)))
            else:
                enc.append((K_INT, int(a)))
        self.ops.append(('call', entry, sidx, enc))

    def sync(self, waiter_raw, signal_raw):
        self.ops.append(('sync', self.streams[waiter_raw], self.streams[signal_raw]))

    def patchable(self, name, entry_name, arg_index):
        """Declare argument `arg_index` (0 = first after the stream) of every recorded call of `entry_name` a run-time
        value `name`: Plan.run(patch={name: int}) overwrites it before the replay (the batch-flip flag of the
        first-layer kernels: one plan serves the mirrored and the plain step)."""
        self.patches.setdefault(name, []).append((L.lib().vqa_plan_opcode(entry_name.encode()), arg_index))

    # ---- layout + encoding
    def _layout(self):
        """Byte offset of every temporary (see the class comment) -> (offsets, arena size)."""
        n_ops = len(self.ops)
        # the op after which stream s's work so far is known to the main stream: the next "main waits for s"
        joins = {}
        for i, o in enumerate(self.ops):
            if o[0] == 'sync' and o[1] == 0:
                joins.setdefault(o[2], []).append(i)
        release = []
        for ptr, nbytes, first, last, streams, exported in self.allocs:
            r = last
            if exported:
                r = n_ops + 1
            else:
                for s in streams:               # (stream indices that touched it)
                    if s != 0:
                        later = [j for j in joins.get(s, ()) if j >= last]
                        r = max(r, later[0] if later else n_ops + 1)
            release.append(r)
        # "stream s waits for the main stream" ordering points: a temporary first touched on s may only take over bytes
        # whose previous owner was done (in main-stream order) before the last such point ahead of that first touch
        forks = {}
        for i, o in enumerate(self.ops):
            if o[0] == 'sync' and o[2] == 0:
                forks.setdefault(o[1], []).append(i)
        free, top, offsets = [], 0, [0] * len(self.allocs)        # free: sorted list of [offset, size, release op]
        live = []                                                  # (release op, offset, size)
        for i, (ptr, nbytes, first, last, streams, exported) in enumerate(self.allocs):
            # blocks whose last use lies before this temporary's creation return to the free list
            still = []
            for r, off, size in live:
                if r < first:
                    free.append([off, size, r])
                else:
                    still.append((r, off, size))
            live = still
            free.sort()
            merged = []
            for off, size, r in free:
                if merged and merged[-1][0] + merged[-1][1] == off:
                    merged[-1][1] += size
                    merged[-1][2] = max(merged[-1][2], r)
                else:
                    merged.append([off, size, r])
            free = merged
            bound = first
            for s_, f_ in streams.items():
                if s_ != 0:
                    before = [j for j in forks.get(s_, ()) if j <= f_]
                    bound = min(bound, before[-1] if before else -1)
            gran = (2 << 20) if nbytes >= (1 << 20) else 512       # torch's allocator: 2 MiB / 512 B granules
            size = -(-max(nbytes, 1) // gran) * gran
            best = None
            for k, (off, fsize, r) in enumerate(free):
                aoff = -(-off // gran) * gran
                if r < bound and aoff + size <= off + fsize and (best is None or fsize < free[best][1]):
                    best = k
            if best is None:
                aoff = -(-top // gran) * gran
                if aoff > top:
                    free.append([top, aoff - top, -1])
                top = aoff + size
            else:
                off, fsize, r = free.pop(best)
                aoff = -(-off // gran) * gran
                if aoff > off:
                    free.append([off, aoff - off, r])
                if aoff + size < off + fsize:
                    free.append([aoff + size, off + fsize - aoff - size, r])
            offsets[i] = aoff
            live.append((release[i], aoff, size))
        return offsets, max(top, ALIGN)

    def encode(self):
        offsets, size = self._layout()

        def word(ref):
            if ref is None:
                return -1
            kind, b, delta = ref
            return (0 << 48) | (offsets[b] + delta) if kind == 'a' else (b << 48) | delta
        words = []
        spots = {name: [] for name in self.patches}     # word indices (in the body) of the patchable values
        for o in self.ops:
            if o[0] == 'sync':
                words += [OP_SYNC, o[1], 1, K_INT, o[2]]
                continue
            _, entry, sidx, enc = o
            words += [entry, sidx, len(enc)]
            for ai, (kind, val) in enumerate(enc):
                for name, targets in self.patches.items():
                    if (entry, ai) in targets:
                        if kind != K_INT:
                            raise PlanError(f'patchable argument {name} is not an integer')
                        spots[name].append(len(words) + 1)
                if kind == K_PTR:
                    words += [K_PTR, word(val)]
                elif kind == K_PTRARR:
                    words += [K_PTRARR, len(val)] + [word(v) for v in val]
                elif kind == K_INTARR:
                    words += [K_INTARR, len(val)] + val
                else:
                    words += [kind, val]
        exports = {name: (offsets[a] + delta, shape, dtype) for name, (a, delta, shape, dtype, _, _) in self.exports.items()}
        live = [(ptr, nbytes, offsets[a] + delta) for (a, delta, _, _, ptr, nbytes) in self.exports.values()]
        self.spots = spots
        return words, size, exports, live

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
        body, size, exports, live = rec.encode()
        words = [MAGIC, len(rec.ops)] + body
        self.n_words = len(words)
        self.words = (ctypes.c_long * self.n_words)(*words)
        self.arena_bytes = size
        self.names = list(rec.names)
        self.stream_names = list(rec.stream_names)
        self.exports = exports
        # the recorded pass's own addresses of what it exported: a later pass (its backward) records against them.  Only
        # the exported tensors - the other temporaries are freed when the recording ends, and their addresses may be
        # handed out again (to the backward's incoming gradient, say)
        self.table = live
        self.n_ops = len(rec.ops)
        self.spots = {name: [i + 2 for i in idx] for name, idx in rec.spots.items()}    # (+ magic, op count)
        self.n_syncs = sum(1 for o in rec.ops if o[0] == 'sync')
        self._events = None         # torch events of the ordering points (made on first use, reused step after step)
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

    def run(self, arena, bases, side_streams=(), patch=None):
        """bases: data pointers (ints) of the external bases in the order they were bound; side_streams: raw handles
        of the named side streams in the order they were declared.  The current stream is stream 0.
        patch: {name: int} for the arguments the recording declared patchable."""
        if patch:
            for name, value in patch.items():
                for i in self.spots[name]:
                    self.words[i] = value
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
        ev = None
        if OWN_EVENTS and self.n_syncs:
            if self._events is None:
                evs = [torch.cuda.Event() for _ in range(self.n_syncs)]
                for e in evs:
                    e.record()                                   # (a torch event has no handle before its first record)
                self._events = (evs, (ctypes.c_void_p * self.n_syncs)(*[e.cuda_event for e in evs]))
            ev = self._events[1]
        status = self._fn(self.words, self.n_words, b, len(self.names), s, len(self.stream_names), ev, self.n_syncs if ev else 0)
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
