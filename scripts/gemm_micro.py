"""GEMM shapes of the training step, timed in isolation (HIP events, 20 reps) + checked against torch fp32."""
import torch
from vqa_amd import ops
from vqa_amd import _lib as L

SHAPES = [  # (trans, M, N, K, label)
    (0, 3584, 3072, 300, 'gi = x W_ih^T'), (0, 256, 3072, 1024, 'gh = h W_hh^T'), (0, 256, 1024, 3072, 'dh = dgh W_hh'),
    (1, 3072, 1024, 3584, 'dW_hh'), (1, 3072, 300, 3584, 'dW_ih'), (0, 3584, 300, 3072, 'dx'),
    (0, 6400, 1024, 256, 'att v-term'), (0, 6400, 256, 1024, 'att dv'), (1, 1024, 256, 6400, 'att dW'),
    (0, 256, 1024, 1024, 'head linear'), (1, 1024, 1024, 256, 'head dW'), (0, 256, 3129, 1024, 'fc'),
    (1, 3129, 1024, 256, 'fc dW'), (0, 256, 1024, 3129, 'fc dx'), (0, 256, 1024, 256, 'att q'), (1, 1024, 256, 256, 'att q dW'),
]
torch.manual_seed(0)
for trans, M, N, K, label in SHAPES:
    Kp = ops.pad8(K)
    if trans:
        A = torch.randn(K, M, device='cuda').bfloat16(); B = torch.randn(K, N, device='cuda').bfloat16()
        ref = A.float().t() @ B.float()
        f = lambda: ops.gemm(A, B, trans=True)
    else:
        A = torch.randn(M, Kp, device='cuda').bfloat16()[:, :K]; B = torch.randn(N, Kp, device='cuda').bfloat16()[:, :K]
        ref = A.float() @ B.float().t()
        f = lambda: ops.gemm(A, B)
    res = []
    for sp in (None, 1, 2, 4, 8, 16):
        f = (lambda sp=sp: ops.gemm(A, B, trans=bool(trans), splits=sp))
        out = f()
        err = float((out.float() - ref).norm() / ref.norm())
        assert err < 1e-5, (label, sp, err)
        for _ in range(3): f()
        L.profile_start()
        for _ in range(8): f()
        prof = L.profile_stop()
        res.append(min(ms for ms, _ in prof['vqa_gemm']) * 1e3)      # device time incl. the split-K reduce
    ms = res[0] / 1e3
    print('%-14s %s %5dx%5dx%5d  auto %6.1f us %6.1f TF/s | splits 1,2,4,8,16: %s' % (label, 'TN' if trans else 'NT', M, N, K, res[0], 2e-9 * M * N * K / ms, ' '.join('%6.1f' % x for x in res[1:])))
