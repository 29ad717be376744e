"""fused GRU step kernels against the GEMM + gate path on the same inputs: python scripts/gru_fused_debug.py H B T"""
import sys, torch
import vqa_amd.gru as G
from vqa_amd import ops
H, B, T = (int(v) for v in sys.argv[1:4])
E = 300
torch.manual_seed(0)
dev = 'cuda'
k = 1 / H ** 0.5
w_ih = torch.empty(3 * H, E, device=dev).uniform_(-k, k); w_hh = torch.empty(3 * H, H, device=dev).uniform_(-k, k)
b_ih = torch.empty(3 * H, device=dev).uniform_(-k, k); b_hh = torch.empty(3 * H, device=dev).uniform_(-k, k)
x = torch.zeros(T * B, ops.pad8(E), device=dev, dtype=torch.bfloat16); x[:, :E] = torch.randn(T * B, E, device=dev)
dh = torch.randn(B, H, device=dev)
res = {}
for fused in (False, True):
    G.FUSED_STEP = fused
    h_f, h_t, save = G._recurrence_fwd(x, E, torch.bfloat16, w_ih, w_hh, b_ih, b_hh, B, T)
    out = G._recurrence_bwd(x, E, torch.bfloat16, h_f, h_t, save, w_ih, w_hh, b_ih, b_hh, dh, None, B, T)
    torch.cuda.synchronize()
    res[fused] = dict(h_f=h_f, save=save, dx=out[0], dw_ih=out[1], dw_hh=out[2], db_ih=out[3], db_hh=out[4])
for k_ in res[True]:
    a, b = res[True][k_].float(), res[False][k_].float()
    print(f'{k_:6s} rel {float((a - b).norm() / b.norm()):.3e}  max {float((a - b).abs().max()):.3e}')
sv = (res[True]['save'] - res[False]['save']).abs().reshape(T, B, 4, H).amax(dim=(1, 3))
print('save diff per (t, field):', sv)
