"""GPU: the one-call path (vqa_run_plan replaying a recorded pass, vqa_amd/plan.py) against the per-launch path it was
recorded from: same launches, same arguments, same streams - so the same bits, in the encoder's output, in every
gradient, in the BatchNorm buffers, with and without the flat gradient buffer, on both precisions."""
import copy
import ctypes

import pytest
import torch

from tests.helpers import build_model, load_case

pytestmark = pytest.mark.gpu


def _run_passes(m, img, q, t, n, flip_seed=None):
    import vqa_model
    crit = vqa_model.SoftScoreBCELoss()
    outs = []
    for it in range(n):
        m.zero_grad(set_to_none=True)
        lo = m(img, q, mirror=bool(it & 1) if flip_seed else False)
        crit(lo, t).backward()
        torch.cuda.synchronize()
        outs.append((lo.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()},
                     {k: b.detach().clone() for k, b in m.named_buffers()}))
    return outs


@pytest.mark.parametrize('case,prec', [('h64_p5_max', 'bf16'), ('h64_p5_max', 'fp32'), ('h512_p0_avg', 'bf16'),
                                       ('h128_p2_max_448', 'bf16')])
def test_replayed_passes_equal_per_launch_passes(case, prec, tmp_path, monkeypatch):
    """Plain autograd (gradients into a block of the pass's own): four passes per-launch, then the same four with
    plans - pass 1 records, 2-4 replay, mirrored on the odd passes (the flip flag of the first-layer kernels is
    patched into the one recorded plan)."""
    from vqa_amd import plan as P
    c = load_case(case)
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    base = build_model(c, tmp_path, prec).cuda().train()
    monkeypatch.setattr(P, 'ENABLED', False)
    ref = _run_passes(copy.deepcopy(base), img, q, t, 4, flip_seed=1)
    monkeypatch.setattr(P, 'ENABLED', True)
    P.clear()
    got = _run_passes(copy.deepcopy(base), img, q, t, 4, flip_seed=1)
    assert len(P._plans) == 2           # forward + backward: the flip flag is a patched argument, not a second plan
    for it, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a[0], b[0]), it
        for k in a[1]:
            assert torch.equal(a[1][k], b[1][k]), (it, k)
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), (it, k)
    nbt = [v for k, v in got[-1][2].items() if k.endswith('num_batches_tracked')]
    assert len(nbt) == 10 and all(int(v) == 4 for v in nbt)


def test_trainer_steps_with_plans_equal_steps_without(tmp_path, monkeypatch):
    """The Trainer (gradients written straight into the flat buffer, side streams, clip + Adamax): five steps with the
    one-call path against five steps launch by launch, parameters and losses bit for bit."""
    from vqa_amd import plan as P
    from vqa_amd.optim import Trainer
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    base = build_model(c, tmp_path, 'bf16').cuda().train()
    res = []
    for enabled in (False, True):
        monkeypatch.setattr(P, 'ENABLED', enabled)
        P.clear()
        m = copy.deepcopy(base)
        tr = Trainer(m)
        torch.manual_seed(3)
        losses = [float(tr.step(img, q, t)) for _ in range(5)]
        res.append((losses, {k: p.detach().clone() for k, p in m.named_parameters()}, float(tr.opt.total_norm)))
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    assert any(k[0] == 'cnn.bwd' for k in P._plans) and any(k[0] == 'cnn.fwd' for k in P._plans)


def test_forward_only_then_backward_records_against_the_arena(tmp_path):
    """A training-mode forward without a backward records only the forward plan; the next pass replays it and its
    backward is then recorded against the replayed pass's arena (views from the plan's exports)."""
    import vqa_model
    from vqa_amd import plan as P
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    m = build_model(c, tmp_path, 'bf16').cuda().train()
    P.clear()
    with torch.no_grad():
        lo0 = m(img, q)
    assert [k[0] for k in P._plans] == ['cnn.fwd']
    m2 = copy.deepcopy(m)
    crit = vqa_model.SoftScoreBCELoss()
    crit(m(img, q), t).backward()                               # forward replayed, backward recorded
    assert sorted(k[0] for k in P._plans) == ['cnn.bwd', 'cnn.fwd']
    P.ENABLED = False
    try:
        crit(m2(img, q), t).backward()
    finally:
        P.ENABLED = True
    for (k, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a.grad, b.grad), k


def test_run_plan_refuses_malformed_plans():
    from vqa_amd import _lib as L, plan as P
    lib = L.lib()
    bases = (ctypes.c_void_p * 1)(0)
    streams = (ctypes.c_void_p * 1)(L._raw_stream())
    bad_magic = (ctypes.c_long * 2)(123, 0)
    assert lib.vqa_run_plan(bad_magic, 2, bases, 1, streams, 1, None, 0) == 1
    empty = (ctypes.c_long * 2)(P.MAGIC, 0)
    assert lib.vqa_run_plan(empty, 2, bases, 1, streams, 1, None, 0) == 0
    truncated = (ctypes.c_long * 3)(P.MAGIC, 1, 0)
    assert lib.vqa_run_plan(truncated, 3, bases, 1, streams, 1, None, 0) == 1
    # an op whose pointer names a base the call does not pass: refused before anything is launched (op 1)
    op = lib.vqa_plan_opcode(b'vqa_zero')
    words = [P.MAGIC, 1, op, 0, 2, P.K_PTR, (5 << 48) | 0, P.K_INT, 16]
    arr = (ctypes.c_long * len(words))(*words)
    assert lib.vqa_run_plan(arr, len(words), bases, 1, streams, 1, None, 0) == (1 | (1 << 8))
    # a well-formed one-op plan: zero 64 bytes at offset 32 of base 0
    x = torch.ones(64, device='cuda')
    bases[0] = x.data_ptr()
    words = [P.MAGIC, 1, op, 0, 2, P.K_PTR, 32, P.K_INT, 64]
    arr = (ctypes.c_long * len(words))(*words)
    assert lib.vqa_run_plan(arr, len(words), bases, 1, streams, 1, None, 0) == 0
    torch.cuda.synchronize()
    assert x[:8].eq(1).all() and x[8:24].eq(0).all() and x[24:].eq(1).all()
