import sys, tempfile, torch
sys.path.insert(0, '.')
import vqa_amd.gru as G
from bench import make_fixture
from vqa_amd import synth
with tempfile.TemporaryDirectory() as tmp:
    qpath, _ = make_fixture(tmp, 10000, 5)
    g = G.GRU(300, 1024, 1, qpath, precision='bf16').cuda()
B, T = 256, 14
q = synth.questions(1, B, T, 10000).cuda()
dh = torch.randn(B, 1024, device='cuda')
for use, fused in ((False, True), (False, False), (True, False), (False, True), (False, False)):
    G.USE_PERSISTENT, G.FUSED_STEP = use, fused
    for it in range(3):
        g.zero_grad()
        torch.cuda.synchronize()
        e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e0.record()
        h = g(q)
        e1.record()
        (h * dh).sum().backward()
        e2.record()
        torch.cuda.synchronize()
    print(f'persistent={use} fused_step={fused}: fwd {e0.elapsed_time(e1):.3f} ms  bwd {e1.elapsed_time(e2):.3f} ms')
