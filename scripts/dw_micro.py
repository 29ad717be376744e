"""Solo timings of the two gradients of a block's second conv: BST data gradient + XF weight gradient against
vqa_conv3x3_dgrad_wgrad.  python scripts/dw_micro.py  (VQA_HIP_LIB picks another build of the library)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_amd import cnn as C

C.FUSE_DGRAD_WGRAD = 32
dt = torch.bfloat16


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, c, H in ((64, 16, 226), (256, 32, 114)):
    torch.manual_seed(0)
    w = torch.randn(c, c, 3, 3) / (3 * c ** 0.5)
    dy = torch.randn(B, H - 2, H - 2, c, device='cuda').to(dt)
    y_lo = (torch.randn(B, H, H, c, device='cuda') * 1.5).to(dt)
    coef = torch.stack([torch.rand(c) + 0.5, torch.randn(c) * 0.4, torch.randn(c) * 0.3, torch.rand(c) + 0.7]).cuda()
    wd = C.pack_weights(w.cuda(), c, dt, dgrad=True)
    dx = torch.empty(B, H, H, c, device='cuda', dtype=dt)
    R, CW, gw = C._wgrad_geom(y_lo, dy)[:3]
    part = torch.empty(gw, c, 9, c, device='cuda')
    t_d = timed(lambda: C.conv_fwd(dy, wd, None, c, 2, False, kind='dgrad', bst=(y_lo, coef), out=dx))
    t_w = timed(lambda: C.conv_wgrad(y_lo, dy, c, 0, in_coeffs=(coef[0], coef[1]), part_out=part))
    t_f = timed(lambda: C.conv_dgrad_wgrad(dy, wd, c, (y_lo, coef), c, out=dx))
    print(f'B={B} {H}x{H} {c}->{c}: dgrad {t_d:.1f} us  wgrad {t_w:.1f} us  sum {t_d + t_w:.1f}  one kernel {t_f:.1f} us', flush=True)
