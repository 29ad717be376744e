"""The first layer's fused weight gradient alone (vqa_conv3x3_first_wgrad_bn at the bench's shape), 20 launches between
two events; VQA_HIP_LIB selects the library build.   python scripts/first_wgrad_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import cnn as M

B = 256
torch.manual_seed(0)
img = (torch.randint(0, 256, (B, 3, 224, 224), device='cuda').float() / 255).half()
y = torch.randn(B, 226, 226, 16, device='cuda').bfloat16()
g = torch.randn(B, 226, 226, 16, device='cuda').bfloat16()
vec = lambda: torch.rand(16, device='cuda') + 0.5
bn = (g, vec(), vec() - 1.0, vec() - 1.0, vec(), vec() * 0.01, vec() * 0.01)
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        M.first_wgrad(img, y, 2, bn=bn)
    e1.record()
    torch.cuda.synchronize()
    print(os.environ.get('VQA_HIP_LIB', 'default').split('/')[-1], 'first_wgrad_bn %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
