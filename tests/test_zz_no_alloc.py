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
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()['allocation.all.allocated']
    free_before = torch.cuda.mem_get_info()[0]                # the DRIVER's free bytes: a hipMalloc inside the library
    L.call('vqa_soft_accuracy_sum_noalloc', logits, target, B, C, logits.stride(0), got)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()['allocation.all.allocated'] == before      # (torch's caching allocator)
    assert torch.cuda.mem_get_info()[0] == free_before                          # (would show here, not above)
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


_CAPTURE = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from vqa_amd import _lib as L, synth
C = 3129
torch.manual_seed(3)
x = torch.randn(300, C, device='cuda')
t = synth.targets(23, 300, C).cuda()
with torch.no_grad():
    x[::2].scatter_(1, t.argmax(1)[::2, None], 30.0)
rows = torch.empty(300, device='cuda')
want = torch.zeros(1, device='cuda')
got = torch.zeros(1, device='cuda')
L.call('vqa_soft_accuracy_sum', x, t, 300, C, C, rows, want)
L.call('vqa_soft_accuracy_sum_noalloc', x, t, 300, C, C, got)       # warm-up OUTSIDE capture: resolves the address
assert float(got) == float(want)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
got.zero_()
with torch.cuda.stream(s):
    g.capture_begin()
    L.call('vqa_soft_accuracy_sum_noalloc', x, t, 300, C, C, got)
    g.capture_end()
torch.cuda.synchronize()
assert float(got) == 0.0                                            # captured, not run
g.replay()
torch.cuda.synchronize()
assert float(got) == float(want) > 0, (float(got), float(want))
x.neg_()                                                            # new data in the same buffers, replayed
L.call('vqa_soft_accuracy_sum', x, t, 300, C, C, rows, want)
g.replay()
torch.cuda.synchronize()
assert float(got) == float(want)
print('CAPTURE_OK')
"""


def test_noalloc_under_graph_capture():
    """The use the header names: the call captured into a HIP graph and replayed (after one warm-up call outside the
    capture, which resolves the scratch address - hipGetSymbolAddress is not a stream operation).  In a child process: a
    capture that goes wrong must not take the test session with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', _CAPTURE, root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'CAPTURE_OK' in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
