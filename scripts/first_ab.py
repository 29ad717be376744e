import sys, torch
from vqa_amd import cnn as M
B=256
img = (torch.randint(0, 256, (B, 3, 224, 224), device='cuda').float() / 255).half()
w = torch.randn(16,3,3,3,device='cuda')*0.1; bias=torch.zeros(16,device='cuda')
for stats in (True, False, True, False):
    for it in range(2):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): M.first_fwd(img,w,bias,2,stats)
        e1.record(); torch.cuda.synchronize()
    print('stats',stats, e0.elapsed_time(e1)/20)
