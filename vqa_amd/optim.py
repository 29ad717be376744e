"""Training-step tail and data parallelism for the drop-in model.

* FlatParams   : all parameters of a module re-pointed into ONE fp32 buffer (and all .grad into another), laid
                 out in reverse execution order (head, GRU, CNN) and cut into buckets.
* GradBuckets  : data-parallel gradient exchange: each bucket is all-reduced (SUM - the loss is a sum,
                 reference vqa_model.py:215) asynchronously on RCCL as soon as autograd has produced its last
                 gradient, so the head/GRU buckets (90 % of the bytes) travel under the CNN backward.
* FlatAdamax   : nn.utils.clip_grad_norm_(max_norm, L2) + torch.optim.Adamax step (reference vqa_model.py:259,
                 :218, :266) as one HIP pass over the flat buffers with no host synchronisation.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


def _bucket_of(name):
    if name.startswith('cnn.'):
        return 2
    if name.startswith('gru.'):
        return 1
    return 0


_layout_ids = iter(range(1, 1 << 62))


class FlatParams:
    def __init__(self, module):
        named = sorted(module.named_parameters(), key=lambda kv: _bucket_of(kv[0]))   # stable: head, gru, cnn
        dev = named[0][1].device
        ALIGN = 64                                              # every parameter starts on a 256-byte boundary
        pad = lambda k: -(-k // ALIGN) * ALIGN
        n = sum(pad(p.numel()) for _, p in named)
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)     # the gaps stay zero in all four buffers
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        # (recorded plans bake a gradient's offset inside this buffer into their pointer arguments: cnn.py keys them by
        # this id, so another buffer - another layout - records plans of its own)
        self.grad._vqa_layout_id = next(_layout_ids)
        self.names, self.params, self.bucket_ranges = [], [], []
        off, cur, start = 0, None, 0
        for name, p in named:
            b = _bucket_of(name)
            if cur is not None and b != cur:
                self.bucket_ranges.append((start, off))
                start = off
            cur = b
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = None
            # the HIP backward writes each gradient into its slot and hands autograd a view of it (ops.grad_dst): no
            # accumulation kernels, no per-step memset of the flat buffer
            p._vqa_grad_dst = self.grad[off:off + k]
            self.names.append(name)
            self.params.append(p)
            off += pad(k)
        self.bucket_ranges.append((start, off))
        self.numel = n

    def zero_grad(self):
        """Gradients are OVERWRITTEN by the next backward (slots never written - the conv biases under a training-mode
        BatchNorm - stay at the initial zero), so "zeroing" is dropping the .grad references."""
        for p in self.params:
            p.grad = None

    def collect(self, params=None):
        """After backward: every p.grad must BE its slot of the flat buffer.  A gradient autograd produced elsewhere
        (a second backward into existing grads, a foreign autograd path) is copied in, so the flat buffer is always
        what the optimizer and the all-reduce see."""
        for p in (self.params if params is None else params):
            dst = p._vqa_grad_dst
            if p.grad is None:
                dst.zero_()
            elif p.grad.data_ptr() != dst.data_ptr():
                dst.copy_(p.grad.reshape(-1))
                p.grad = dst.view(p.shape)


# Data parallelism keeps the single-GPU stream layout: the trainer's stream, the weight-gradient stream, the question
# encoder's stream, and RCCL's own - four streams on the four hardware queues a process has (GPU_MAX_HW_QUEUES; the
# default stream is idle during a step; profiles/r03_dp_queues.json shows the four on distinct queue ids).
# VQA_DP_FOLD_STREAMS=1 puts the question encoder on the weight-gradient stream instead (two compute streams + RCCL's,
# for a runtime with fewer queues): same-box A/B with one rank, bench.py --force-buckets, 4.16-4.23 ms/step plain,
# 4.25-4.28 with the buckets armed, 4.46 folded - the recurrence's BPTT then queues in front of the conv weight
# gradients and the end-of-backward join waits for both.
DP_FOLD_STREAMS = os.environ.get('VQA_DP_FOLD_STREAMS', '0') != '0'
# A bucket whose last gradients are produced on a side stream (the head's parameter gradients) is all-reduced FROM that
# stream; VQA_DP_ISSUE_FROM_PRODUCER=0: the hook's stream (the critical chain) waits for the side stream instead.
DP_ISSUE_FROM_PRODUCER = os.environ.get('VQA_DP_ISSUE_FROM_PRODUCER', '1') != '0'


class GradBuckets:
    """Bucketed asynchronous gradient all-reduce (SUM) over torch.distributed (backend 'nccl' = RCCL on ROCm,
    'gloo' in the CPU tests)."""

    def __init__(self, flat: FlatParams, group=None, force=False, all_reduce=None):
        """force=True arms the hooks and the all-reduces even in a 1-rank group (tests: the real RCCL path on one GPU).
        all_reduce (tests): a callable buf -> handle with .wait() that stands in for the asynchronous
        dist.all_reduce(buf, SUM) - with it the hooks are armed without a process group."""
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and (dist.is_initialized() or all_reduce is not None))
        self._all_reduce = all_reduce or (lambda buf: dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group,
                                                                      async_op=True))
        self.pending, self.handles = [], []
        self._left = []
        if self.active:
            # a bucket is ready when the LAST of its parameters has accumulated its gradient
            ends = [e for _, e in flat.bucket_ranges]
            self._bucket_idx = []
            for p in flat.params:
                off = p._vqa_grad_dst.storage_offset()
                self._bucket_idx.append(next(i for i, e in enumerate(ends) if off < e))
            for i, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(self._bucket_idx[i]))
        self.reset()

    def reset(self):
        self.handles = []
        counts = [0] * len(self.flat.bucket_ranges)
        if self.active:
            for b in self._bucket_idx:
                counts[b] += 1
        self._left = counts

    def _launch(self, b):
        """All-reduce bucket b.  Its gradients were written by kernels on the current stream, or by work a backward
        deferred to a side stream (ops.defer_join names what that work writes).  In the second case the collective is
        issued with THAT stream current - torch's process group orders its own stream behind the current one - after
        that stream has been ordered behind this one; the current stream (the critical chain: the encoders' backward
        comes next) does not wait for the side stream's kernels."""
        s, e = self.flat.bucket_ranges[b]
        buf = self.flat.grad[s:e]
        lo = buf.data_ptr()
        hi = lo + 4 * (e - s)
        side = ops.deferred_stream(lo, hi) if DP_ISSUE_FROM_PRODUCER else None
        if side is None:
            ops.join_deferred(lo, hi)
        self.flat.collect([q for q, bq in zip(self.flat.params, self._bucket_idx) if bq == b])
        if side is None:
            self.handles.append(self._all_reduce(buf))
            return
        side.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(side):
            self.handles.append(self._all_reduce(buf))

    def _make_hook(self, b):
        def hook(_p):
            self._left[b] -= 1
            if self._left[b] == 0:
                self._launch(b)
        hook._vqa_join_aware = True         # (_HeadFunction.backward: this hook orders itself behind the side stream)
        return hook

    def wait(self):
        """End of backward: every bucket must have been launched.  A bucket whose countdown did not reach zero holds a
        parameter that received no gradient in this pass (an unused branch): its hook never fired, so the bucket is
        reduced here - zero-filled slots for the missing gradients - instead of silently leaving the ranks with
        different gradients."""
        if self.active:
            for b, left in enumerate(self._left):
                if left != 0:
                    self._launch(b)
        for h in self.handles:
            h.wait()
        self.reset()


class FlatAdamax:
    def __init__(self, flat: FlatParams, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=0.25):
        self.flat, self.lr, self.betas, self.eps, self.max_norm = flat, lr, betas, eps, max_norm
        dev = flat.flat.device
        self.m = torch.zeros_like(flat.flat)
        self.u = torch.zeros_like(flat.flat)
        self.step_count = 0
        self.part = torch.empty(1024, device=dev, dtype=torch.float64)
        self.sq = torch.zeros(1, device=dev)
        self.coef = torch.ones(1, device=dev)
        self.total_norm = torch.zeros(1, device=dev)     # device scalar: read it only when you want to sync

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self):
        self.step_count += 1
        f = self.flat
        L.check(L.lib().vqa_clip_adamax(L.stream(), L.ptr(f.flat), L.ptr(f.grad), L.ptr(self.m), L.ptr(self.u),
                                        ctypes.c_long(f.numel), ctypes.c_float(self.max_norm or 0.0),
                                        ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]),
                                        ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps),
                                        ctypes.c_int(self.step_count), L.ptr(self.part), L.ptr(self.sq),
                                        L.ptr(self.coef), L.ptr(self.total_norm)), 'vqa_clip_adamax')


STEP_PRIORITY = os.environ.get('VQA_STEP_PRIORITY', '1') != '0'
# the loss formed in the classifier GEMM's epilogue (VQA.forward_loss) instead of a criterion pass over the logits
FUSED_LOSS = os.environ.get('VQA_FUSED_LOSS', '1') != '0'
# The host may run at most this many steps ahead of the GPU (0 = unbounded): before it issues step n it waits for step
# n - MAX_AHEAD to have finished.  With the encoder passes issued by one call each the host needs 2.5 ms per 4.1 ms step
# and, unbounded, ends up tens of steps ahead - and the GPU gets SLOWER for it (same-box A/B, 3 alternations of 60
# steps: unbounded 4.16 ms/step, at most 4 ahead 4.10, at most 2 ahead 4.09, at most 1 ahead 4.33 - the host then
# cannot hide its own issue time).
MAX_AHEAD = int(os.environ.get('VQA_MAX_AHEAD', '2'))
_STEP_STREAMS = {}


class Trainer:
    """One reference training step (vqa_model.py:248-266) on the HIP path:
    zero_grad -> flip -> forward -> sum-BCE -> backward (-> bucketed all-reduce) -> clip 0.25 -> Adamax."""

    def __init__(self, model, lr=2e-3, max_norm=0.25, force_buckets=False, all_reduce=None):
        from .vqa_model import SoftScoreBCELoss
        self.model = model
        self.flat = FlatParams(model)
        self.buckets = GradBuckets(self.flat, force=force_buckets, all_reduce=all_reduce)
        if self.buckets.active and DP_FOLD_STREAMS:
            model._dp_fold_streams = True           # (a plain bool: the module still pickles whole)
        self.opt = FlatAdamax(self.flat, lr=lr, max_norm=max_norm)
        self.criterion = SoftScoreBCELoss()
        self._marks, self._mark_i = [], 0           # one event per step in flight (a ring of MAX_AHEAD)
        self._one = torch.ones((), device=self.flat.flat.device)

    def step(self, images, questions, target, flip=True, dropout_noise=None):
        """The step runs on a high-priority HIP stream of its own, ordered after and before the caller's current
        stream: its kernels are the critical chain, and the work the model puts on side streams (question encoder,
        weight gradients: normal priority) then only fills what that chain leaves idle (4.96 -> 4.90 ms/step)."""
        if not STEP_PRIORITY or not images.is_cuda:
            return self._step(images, questions, target, flip, dropout_noise)
        cur = torch.cuda.current_stream(images.device)
        hp = self.stream
        if cur == hp:                                  # the caller already runs its loop on the trainer's stream
            return self._step(images, questions, target, flip, dropout_noise)
        hp.wait_stream(cur)
        with torch.cuda.stream(hp):
            loss = self._step(images, questions, target, flip, dropout_noise)
        cur.wait_stream(hp)
        loss.record_stream(cur)
        return loss

    @property
    def stream(self):
        """The trainer's high-priority stream.  A training loop that runs under `with torch.cuda.stream(trainer.stream)`
        (after `trainer.stream.wait_stream(torch.cuda.current_stream())` for tensors made elsewhere) saves the two
        cross-stream hand-overs per step that step() otherwise makes to keep the caller's stream semantics."""
        dev = self.flat.flat.device
        key = str(dev)
        if key not in _STEP_STREAMS:
            _STEP_STREAMS[key] = torch.cuda.Stream(dev, priority=-1)
        return _STEP_STREAMS[key]

    def _throttle(self):
        """Before issuing a step: wait (on the host) for the step MAX_AHEAD steps back."""
        if MAX_AHEAD > 0 and len(self._marks) >= MAX_AHEAD:
            self._marks[(self._mark_i + 1) % len(self._marks)].synchronize()     # the oldest mark of the ring

    def _mark(self):
        if MAX_AHEAD > 0:
            if len(self._marks) < MAX_AHEAD:
                self._marks.append(torch.cuda.Event())
                self._mark_i = len(self._marks) - 1
            else:
                self._mark_i = (self._mark_i + 1) % len(self._marks)
            self._marks[self._mark_i].record()

    def _step(self, images, questions, target, flip=True, dropout_noise=None):
        self._throttle()
        self.opt.zero_grad()
        # the reference flips the batch tensor (vqa_model.py:250); same coin, but the first conv reads it mirrored
        mirror = flip and self.model.flip.draw()
        if FUSED_LOSS:
            _, loss = self.model.forward_loss(images, questions, target, mirror=mirror, dropout_noise=dropout_noise)
        else:
            logits = self.model(images, questions, mirror=mirror, dropout_noise=dropout_noise)
            loss = self.criterion(logits, target)
        # The step is host-bound (~150 launches): with autograd's worker thread off, the backward's Python code runs
        # on this thread - no hand-over per node (issue time 4.17 -> 3.72 ms per step)
        # (the root gradient is a resident 1.0: loss.backward() alone would fill a fresh one every step)
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward(self._one)
        ops.join_deferred()
        if not self.buckets.active:
            self.flat.collect()
        self.buckets.wait()
        self.opt.step()
        self._mark()
        return loss.detach()
