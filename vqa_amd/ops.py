"""Thin Python wrappers over the C-ABI entry points shared by the GRU, the fusion head, the loss and the optimizer.
Every function launches hand-written HIP kernels on torch's current stream; nothing here computes on the CPU."""
import ctypes

import torch

from . import _lib as L


def pad8(n):
    return (n + 7) // 8 * 8


def gemm(A, B, trans=False, M=None, N=None, K=None, out=None, out_dtype=torch.float32, bias=None, rowadd=None,
         group=1, act=0, accumulate=False, c2=None, splits=None, emul=None, emode=0, emul2=None):
    """NT: out[M,N] = A[M,K] @ B[N,K]^T.   TN (trans=True): out[M,N] = A[K,M]^T @ B[K,N].
    A / B may be row-strided views (stride(1) == 1).  Epilogue: + bias[n] + rowadd[m // group, n], ReLU (act=1),
    accumulate into `out`, optional bf16 copy `c2`; emul / emode: vqa_gemm's epilogue multiplier (emode 1:
    c2 = bf16(out * emul); emode 16 + act: out *= act'(emul), the derivative from the activation's output; emode 32 +
    act: out = x * emul2 * act'(emul), c2 = bf16(x * emul * act'(emul2)), emul2 with emul's row stride)."""
    assert A.stride(1) == 1 and B.stride(1) == 1 and A.dtype == B.dtype
    if trans:
        K = A.shape[0] if K is None else K
        M = A.shape[1] if M is None else M
        N = B.shape[1] if N is None else N
    else:
        M = A.shape[0] if M is None else M
        K = A.shape[1] if K is None else K
        N = B.shape[0] if N is None else N
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=out_dtype)
    assert out.stride(1) == 1
    if splits is None:
        splits = gemm_splits(M, N, K, A.dtype, trans)
    ws = torch.empty(splits, M, N, device=A.device, dtype=torch.float32) if splits > 1 else None
    tag = lambda: dict(kind='tn' if trans else 'nt', shape=f'{M}x{N}x{K}', flops=2.0 * M * N * K,
               bytes=float((M * K + N * K) * A.element_size() + M * N * out.element_size()))
    L.call('vqa_gemm', L.dt(A), int(trans), M, N, K, A, A.stride(0), B, B.stride(0), out, out.stride(0), L.dt(out),
           bias, rowadd, group, act, int(accumulate), c2, c2.stride(0) if c2 is not None else 0, ws, splits,
           emul, emul.stride(0) if emul is not None else 0, emode, emul2, tag=tag)
    return out


def gemm_splits(M, N, K, dtype, trans=False):
    """split-K when the 64x64 output grid alone cannot fill 256 CUs (2 workgroups each) and the reduction is long;
    thresholds from scripts/gemm_micro.py + rocprofv3 on the step's shapes"""
    blocks = -(-M // 64) * -(-N // 64)
    if not trans and dtype == torch.bfloat16 and -(-M // 32) * -(-N // 32) <= 256 and K <= 1024:
        return 1              # gemm.hip: 32x32 tiles with eight stages of loads in flight, no split-K pass
    if blocks >= 160:
        # 128x128 tiles (gemm.hip picks them from 160 tiles on): a grid that leaves a quarter of the CUs without a
        # workgroup and a long reduction run faster as two K slices, two workgroups per CU (the recurrence's dW_hh)
        big = -(-M // 128) * -(-N // 128)
        if 160 <= big < 256 and K >= 2048:
            return 2
        # (NT: 3584 x 300 x 3072 is 35 us whole and 42 in two slices)
        return 2 if (trans and K >= 3072 and blocks < 384) else 1
    return max(1, min(K // 384, 320 // blocks, 16))


_scratch = {}


def scratch(dev):
    """256 doubles + a few floats of per-device reduction scratch (stream-ordered use only)."""
    key = str(dev)
    if key not in _scratch:
        _scratch[key] = torch.empty(256, device=dev, dtype=torch.float64)
    return _scratch[key]


def grad_dst(p):
    """A fresh view of parameter p's slot in the flat gradient buffer (optim.FlatParams sets `_vqa_grad_dst`), or
    None.  A backward that writes its result there and RETURNS this view lets autograd adopt it as p.grad without a
    copy or an accumulation kernel; that is only valid while p.grad is None (nothing to accumulate into)."""
    d = getattr(p, '_vqa_grad_dst', None)
    if d is None or p.grad is not None:
        return None
    return d.view(p.shape)


# ---- work deferred to a side stream inside a backward (head / conv weight gradients): whoever reads its results on
# another stream joins first - the end-of-backward callback below, the data-parallel bucket hooks, Trainer.step.
# Keyed by DEVICE and guarded by a lock: the events of one GPU are one list whatever model or thread recorded them (a
# backward's node functions may run on an autograd worker thread while its final callbacks run on the thread that called
# backward(), so a per-thread list would lose them).  Nothing is ever dropped while it may still be running: a reset
# only prunes events that have completed, and a join waits for every live event of the device - joining another model's
# deferred work as well costs a wait, never a wrong number.
import threading

_deferred = {}                      # device index -> [(event, stream, [(lo, hi) byte ranges written] or None)]
_deferred_lock = threading.Lock()


def _dev_key(stream=None):
    return stream.device.index if stream is not None else torch.cuda.current_device()


def defer_reset():
    """Start of a backward that may defer work: forget the events of earlier passes that have completed."""
    key = _dev_key()
    with _deferred_lock:
        _deferred[key] = [e for e in _deferred.get(key, ()) if not e[0].query()]


def defer_join(side, outputs=None):
    """Call at the end of side-stream work issued from inside a backward: records its completion and arranges for the
    stream that called .backward() to wait for it before backward() returns (autograd runs its final callbacks on
    that stream).  outputs: the tensors that work writes (so that a reader of some other memory need not wait)."""
    spans = None if outputs is None else [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in outputs]
    with _deferred_lock:
        _deferred.setdefault(_dev_key(side), []).append((side.record_event(), side, spans))
    torch.autograd.Variable._execution_engine.queue_callback(join_deferred)


def _touches(spans, lo, hi):
    return spans is None or lo is None or any(a < hi and lo < b for a, b in spans)


def join_deferred(lo=None, hi=None):
    """Make the current stream wait for the side-stream work deferred on this device (idempotent); with a byte range
    [lo, hi) only for the work that writes into it (work recorded without `outputs` always counts)."""
    if not _deferred:               # nothing was ever deferred in this process (e.g. the CPU tests of the bucket logic)
        return
    with _deferred_lock:
        live = [e for e in _deferred.get(_dev_key(), ()) if _touches(e[2], lo, hi)]
    if live:
        cur = torch.cuda.current_stream()
        for ev, _, _ in live:
            cur.wait_event(ev)


def deferred_stream(lo, hi):
    """The side stream on which the most recently deferred, still recorded work that writes into [lo, hi) runs (None
    if there is none): a collective over those results is issued FROM that stream - ordered behind the work by the
    stream itself - so the stream that runs the rest of the backward does not wait for it (optim.GradBuckets)."""
    if not _deferred:
        return None
    with _deferred_lock:
        for ev, stream, spans in reversed(_deferred.get(_dev_key(), ())):
            if spans is not None and _touches(spans, lo, hi):
                return stream
    return None


def dot(a, b):
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    L.check(L.lib().vqa_dot(L.stream(), L.ptr(a), L.ptr(b), ctypes.c_long(a.numel()), L.ptr(scratch(a.device)),
                            L.ptr(out)), 'vqa_dot')
    return out


def colsum(x, M=None, N=None, out=None):
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    # short matrices (the head's [B, H] gradients): one launch, 16 row lanes per block; tall ones (the recurrence's
    # [T*B, 3H] gate gradients): up to 64 row slabs + the fixed-order combine
    slabs = 1 if M <= 1024 else min(64, M // 64)
    scr = torch.empty(slabs, N, device=x.device, dtype=torch.float32) if slabs > 1 else None
    L.call('vqa_colsum', L.dt(x), x, M, N, x.stride(0), out, scr, slabs)
    return out


def cast_pad(src, dtype, ldd=None, scale=None):
    """[M,N] -> contiguous [M,ldd] in `dtype`, zero-padded columns; returns the [M,N] view.
    scale: device scalar multiplied in (the incoming gradient of a scalar loss, no host read)."""
    M, N = src.shape
    ldd = N if ldd is None else ldd
    if src.dtype == dtype and ldd == src.stride(0) and scale is None:
        return src
    dst = torch.empty(M, ldd, device=src.device, dtype=dtype)
    L.call('vqa_cast_pad', L.dt(src), L.dt(dst), src, M, N, src.stride(0), dst, ldd, scale)
    return dst[:, :N]


def cast_transpose(src, dtype):
    """f32 [M,N] -> its transpose [N,M] in `dtype` (row stride padded to 8, pad columns zero); returns the [N,M] view."""
    M, N = src.shape
    assert src.dtype == torch.float32 and src.stride(1) == 1
    dst = torch.empty(N, pad8(M), device=src.device, dtype=dtype)
    L.call('vqa_cast_transpose', L.dt(dst), src, M, N, src.stride(0), dst, dst.stride(0))
    return dst[:, :M]


def fc_bce(x, W, target):
    """The classifier GEMM with the loss in its epilogue (reference vqa_model.py:103 + :215): x [B,H] and W [C,H] in the
    MFMA dtype, target dense fp32 [B,C] or the sparse pair (labels, scores) -> (logits fp32 [B,C], loss [1],
    dlogits = sigmoid(logits) - target)."""
    B, H = x.shape
    C = W.shape[0]
    dev = x.device
    logits = torch.empty(B, C, device=dev, dtype=torch.float32)
    dl = torch.empty(B, C, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    part = torch.empty(L.lib().vqa_fc_bce_part(B, C), device=dev, dtype=torch.float64)
    sp = _sparse(target)
    if sp is not None:
        dense, ldt, labels, scores, nl = None, 0, sp[0], sp[1], sp[0].shape[1]
    else:
        dense = target.contiguous()
        assert dense.dtype == torch.float32 and tuple(dense.shape) == (B, C)
        ldt, labels, scores, nl = C, None, None, 0
    L.call('vqa_fc_bce', L.dt(x), B, C, H, x, x.stride(0), W, W.stride(0), logits, C, dense, ldt, labels, scores, nl,
           dl, C, part, loss)
    return logits, loss, dl


def weight_norm_fwd(v, g, dtype, transpose=True):
    """weight_norm(dim=None) (reference vqa_model.py:59-75): W = g * v / ||v||_F.
    Returns (W [out,in] view with padded row stride, W^T [in,out] view or None, sumsq)."""
    out_f, in_f = v.shape
    sumsq = dot(v, v)
    w = torch.empty(out_f, pad8(in_f), device=v.device, dtype=dtype)
    wt = torch.empty(in_f, pad8(out_f), device=v.device, dtype=dtype) if transpose else None
    L.call('vqa_wn_apply', L.dt(w), v, g, sumsq, out_f, in_f, w, w.stride(0), wt, wt.stride(0) if transpose else 0)
    return w[:, :in_f], (wt[:, :out_f] if transpose else None), sumsq


def weight_norm_bwd(dw, v, g, sumsq):
    """dW (fp32 [out,in] contiguous) -> (dg scalar, dv); written into the flat gradient slots when there are any."""
    d = dot(dw, v)
    dv, dg = grad_dst(v), grad_dst(g)
    if dv is None:
        dv = torch.empty_like(v)
    if dg is None:
        dg = torch.empty((), device=v.device, dtype=torch.float32)
    L.check(L.lib().vqa_wn_bwd(L.stream(), L.ptr(dw), L.ptr(v), L.ptr(g), L.ptr(sumsq), L.ptr(d),
                               ctypes.c_long(v.numel()), L.ptr(dv), L.ptr(dg)), 'vqa_wn_bwd')
    return dg, dv


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def weight_norm_fwd_all(layers, dtype):
    """layers = [(v, g, transpose)]: weight_norm_fwd of all of them in two launches.
    Returns ([(W, W^T or None)], sumsq [n] - kept for weight_norm_bwd_all)."""
    n, dev = len(layers), layers[0][0].device
    ws, wts, dims = [], [], []
    for v, g, tr in layers:
        out_f, in_f = v.shape
        w = torch.empty(out_f, pad8(in_f), device=dev, dtype=dtype)
        wt = torch.empty(in_f, pad8(out_f), device=dev, dtype=dtype) if tr else None
        ws.append(w)
        wts.append(wt)
        dims += [out_f, in_f, w.stride(0), wt.stride(0) if tr else 0]
    sumsq = torch.empty(n, device=dev, dtype=torch.float32)
    part = torch.empty(n * 256, device=dev, dtype=torch.float64)
    L.check(L.lib().vqa_wn_fwd_all(L.stream(), L.dt(ws[0]), n, _ptrs([l[0] for l in layers]),
                                   _ptrs([l[1] for l in layers]), _ptrs(ws), _ptrs(wts),
                                   (ctypes.c_int * (4 * n))(*dims), L.ptr(sumsq), L.ptr(part)), 'vqa_wn_fwd_all')
    views = [(w[:, :l[0].shape[1]], None if wt is None else wt[:, :l[0].shape[0]])
             for l, w, wt in zip(layers, ws, wts)]
    return views, sumsq


def weight_norm_bwd_all(items, sumsq):
    """items = [(dW fp32 contiguous, v, g)] in the order of weight_norm_fwd_all -> [(dg, dv)], written into the flat
    gradient slots when there are any."""
    n, dev = len(items), items[0][1].device
    dvs, dgs, dims = [], [], []
    for dw, v, g in items:
        assert dw.is_contiguous() and dw.dtype == torch.float32 and dw.numel() == v.numel()
        dv, dg = grad_dst(v), grad_dst(g)
        dvs.append(torch.empty_like(v) if dv is None else dv)
        dgs.append(torch.empty((), device=dev, dtype=torch.float32) if dg is None else dg)
        dims += [v.shape[0], v.shape[1]]
    part = torch.empty(n * 256, device=dev, dtype=torch.float64)
    L.check(L.lib().vqa_wn_bwd_all(L.stream(), n, _ptrs([i[0] for i in items]), _ptrs([i[1] for i in items]),
                                   _ptrs([i[2] for i in items]), _ptrs(dvs), _ptrs(dgs),
                                   (ctypes.c_int * (2 * n))(*dims), L.ptr(sumsq), L.ptr(part)), 'vqa_wn_bwd_all')
    return list(zip(dgs, dvs))


def bce_sum(logits, target, want_grad=True):
    """sum-reduced BCE-with-logits (reference vqa_model.py:215) -> (loss [1], dlogits or None)."""
    B, C = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dl = torch.empty(B, C, device=logits.device, dtype=torch.float32) if want_grad else None
    L.call('vqa_bce_sum', logits, target, B, C, logits.stride(0), dl, scratch(logits.device), loss)
    return loss, dl


def _sparse(target):
    """(labels int64 [B,L], scores fp32 [B,L]) if `target` is the sparse form, else None."""
    if isinstance(target, (tuple, list)) and len(target) == 2:
        labels, scores = target
        assert labels.dtype == torch.int64 and labels.shape == scores.shape and labels.dim() == 2
        return labels.contiguous(), scores.float().contiguous()
    return None


def bce_sum_sparse(logits, labels, scores, want_grad=True):
    """bce_sum with sparse soft-score targets: row b has the (label, score) slots labels[b], scores[b]; label < 0 is
    an empty slot; labels are distinct within a row (compute_softscore.py:192-239 produces them so)."""
    B, C = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dl = torch.empty(B, C, device=logits.device, dtype=torch.float32) if want_grad else None
    rowloss = torch.empty(B, device=logits.device, dtype=torch.float64)
    L.call('vqa_bce_sum_sparse', logits, labels, scores, B, C, labels.shape[1], logits.stride(0), dl, rowloss, loss)
    return loss, dl


def soft_accuracy_sum(logits, target):
    """sum_b target[b, argmax logits[b]] on the device (reference vqa_model.py:127-130) -> tensor [1]."""
    B, C = logits.shape
    rows = torch.empty(B, device=logits.device, dtype=torch.float32)
    acc = torch.empty(1, device=logits.device, dtype=torch.float32)
    sp = _sparse(target)
    if sp is not None:
        L.call('vqa_soft_accuracy_sum_sparse', logits, sp[0], sp[1], B, C, sp[0].shape[1], logits.stride(0), rows, acc)
        return acc
    L.call('vqa_soft_accuracy_sum', logits, target.contiguous(), B, C, logits.stride(0), rows, acc)
    return acc
