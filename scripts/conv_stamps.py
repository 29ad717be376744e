"""Phase timing inside conv3x3_igemm_tile (needs the -DVQA_STAMPS build: VQA_HIP_LIB=.../libvqa_hip_stamps.so).
usage: conv_stamps.py <layer> <fwd|dgrad>     prints mean cycles between the stamps of wave 0 over all workgroups"""
import ctypes
import sys
import numpy as np
import torch
from vqa_amd import cnn as M, _lib as L
li, kind = int(sys.argv[1]), sys.argv[2]
B = 256
sizes = [226, 224, 114, 112, 58, 56, 30, 28, 12, 10]
pads = [2, 0, 2, 0, 2, 0, 2, 0, 0, 0]
cin, cout = max(M.CHANNELS[li], 16), M.CHANNELS[li + 1]
OH = sizes[li]; IH = OH + 2 - 2 * pads[li]
dt = torch.bfloat16
x = torch.randn(B, IH, IH, cin, device='cuda').to(dt)
w = torch.randn(cout, M.CHANNELS[li], 3, 3, device='cuda') * 0.1
dy = torch.randn(B, OH, OH, cout, device='cuda').to(dt)
bias = torch.zeros(cout, device='cuda')
wpk = M.pack_weights(w, cin, dt, False); wpk_d = M.pack_weights(w, cin, dt, True)
for _ in range(3):
    if kind == 'fwd':
        M.conv_fwd(x, wpk, bias, cout, pads[li], True)
    else:
        M.conv_fwd(dy, wpk_d, None, cin, 2 - pads[li], False, kind='dgrad')
torch.cuda.synchronize()
buf = np.zeros((16384, 12), dtype=np.uint64)
assert L.lib().dbg_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf[buf[:, 8] > 0].astype(np.int64)
names = ['decode+bias', 'fetch0 issue+stash0 (wait loads)', 'barrier', 'fetch1 issue + MFMA chunk0', 'barrier', 'chunks 1..', 'epilogue math+transpose', 'stats+row stores']
d = np.diff(s[:, :9], axis=1)
print(f'layer {li} {kind}: {len(s)} workgroups, total {d.sum(1).mean():.0f} cycles (100 MHz counter ticks?)')
for n, v in zip(names, d.mean(0)):
    print(f'  {n:40s} {v:9.0f}  {100 * v / d.sum(1).mean():5.1f} %')
