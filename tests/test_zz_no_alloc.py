"""GPU: the one entry point that allocates nothing (vqa_soft_accuracy_sum_noalloc, include/vqa_hip.h) against the
workspace-taking entry point it is a variant of, through the C-ABI: bit-identical on two calls in a row, the second
with a smaller input (the static scratch keeps the first call's rows), on strided logits, and refusing a batch beyond
its capacity without launching anything."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(logits, target):
    from vqa_amd import _lib as L
    B, C = logits.shape
    rows = torch.empty(B, device='cuda', dtype=torch.float32)
    want = torch.full((1,), -1.0, device='cuda')
    got = torch.full((1,), -2.0, device='cuda')
    L.call('vqa_soft_accuracy_sum', logits, target, B, C, logits.stride(0), rows, want)
    before = torch.cuda.memory_stats()['allocation.all.allocated']
    L.call('vqa_soft_accuracy_sum_noalloc', logits, target, B, C, logits.stride(0), got)
    assert torch.cuda.memory_stats()['allocation.all.allocated'] == before
    return float(want), float(got)


def test_noalloc_matches_workspace_variant_twice():
    from vqa_amd import synth
    C = 3129
    torch.manual_seed(11)
    big = torch.randn(1024, C, device='cuda')
    t_big = synth.targets(21, 1024, C).cuda()
    with torch.no_grad():                                     # make the arg-max hit a scored label now and then
        lab = t_big.argmax(1)
        big[::3, :].scatter_(1, lab[::3, None], 30.0)
    want, got = _both(big, t_big)
    assert want == got and want > 0
    small = torch.randn(37, C + 7, device='cuda')[:, :C]      # smaller, and a row stride != C
    t_small = synth.targets(22, 37, C).cuda()
    with torch.no_grad():
        small.scatter_(1, t_small.argmax(1)[:, None], 30.0)
    want2, got2 = _both(small, t_small)
    assert want2 == got2 and want2 > 0 and want2 != want
    want3, got3 = _both(big, t_big)                           # and back up again
    assert (want3, got3) == (want, got)


def test_noalloc_rejects_over_capacity():
    from vqa_amd import _lib as L
    x = torch.zeros(4, 8, device='cuda')
    acc = torch.full((1,), 5.0, device='cuda')
    fn = L.lib().vqa_soft_accuracy_sum_noalloc
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    status = fn(L.stream(), x.data_ptr(), x.data_ptr(), 65537, 8, 8, acc.data_ptr())
    assert status == 3
    torch.cuda.synchronize()
    assert float(acc) == 5.0                                  # nothing was launched
    with pytest.raises(L.VqaHipError):
        L.call('vqa_soft_accuracy_sum_noalloc', x, x, 0, 8, 8, acc)
