"""Micro-benchmark of single conv layers (for rocprofv3 --pmc runs).  usage: conv_micro.py <layer> <kind> [reps]"""
import sys
import torch
from vqa_amd import cnn as M
li, kind = int(sys.argv[1]), sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = 256
sizes = [226, 224, 114, 112, 58, 56, 30, 28, 12, 10]          # output sizes
pads = [2, 0, 2, 0, 2, 0, 2, 0, 0, 0]
cin, cout = max(M.CHANNELS[li], 16), M.CHANNELS[li + 1]
OH = sizes[li]
IH = OH + 2 - 2 * pads[li]
dt = torch.bfloat16
x = torch.randn(B, IH, IH, cin, device='cuda').to(dt)
w = torch.randn(cout, M.CHANNELS[li], 3, 3, device='cuda') * 0.1
dy = torch.randn(B, OH, OH, cout, device='cuda').to(dt)
bias = torch.zeros(cout, device='cuda')
wpk = M.pack_weights(w, cin, dt, False)
wpk_d = M.pack_weights(w, cin, dt, True)
img = (torch.randint(0, 256, (B, 3, 224, 224), device='cuda').float() / 255).half()
by = torch.randn(B, IH, IH, cin, device='cuda').to(dt)
bcoef = torch.rand(4, cin, device='cuda') + 0.5
isc = torch.rand(cin, device='cuda') + 0.5
ish = torch.randn(cin, device='cuda') * 0.1
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(2):
    e0.record()
    for _ in range(reps):
        if kind == 'first':
            M.first_fwd(img, w, bias, pads[0], True)
        elif kind == 'first_wgrad':
            M.first_wgrad(img, dy, pads[0])
        elif kind == 'fwdxf':
            M.conv_fwd(x, wpk, bias, cout, pads[li], True, in_coeffs=(isc, ish))
        elif kind == 'wgradxf':
            M.conv_wgrad(x, dy, M.CHANNELS[li], pads[li], in_coeffs=(isc, ish))
        elif kind == 'fwd':
            M.conv_fwd(x, wpk, bias, cout, pads[li], True)
        elif kind == 'dgradbst':
            M.conv_fwd(dy, wpk_d, None, cin, 2 - pads[li], False, kind='dgrad', bst=(by, bcoef))
        elif kind == 'dgrad':
            M.conv_fwd(dy, wpk_d, None, cin, 2 - pads[li], False, kind='dgrad')
        else:
            M.conv_wgrad(x, dy, M.CHANNELS[li], pads[li])
    e1.record()
    torch.cuda.synchronize()
print(f'layer {li} {kind}: {e0.elapsed_time(e1) / reps:.4f} ms')
