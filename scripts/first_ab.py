"""The first layer's forward kernel alone at the bench's shape, 20 launches between two events, per setting of
VQA_FIRST_RESERVE (persistent workgroups per CU = 5 - reserve).   python scripts/first_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import cnn as M
B = 256
img = (torch.randint(0, 256, (B, 3, 224, 224), device='cuda').float() / 255).half()
w = torch.randn(16, 3, 3, 3, device='cuda') * 0.1
bias = torch.zeros(16, device='cuda')
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        M.first_fwd(img, w, bias, 2, True)
    e1.record()
    torch.cuda.synchronize()
    if rep >= 3:
        print('reserve', M.FIRST_RESERVE, 'first_fwd %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
