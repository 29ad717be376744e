"""GPU: MFMA GEMM (vqa_gemm) and layout probes against plain PyTorch fp32 references, through the C-ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(A, B, trans, out_dtype=torch.float32, bias=None, rowadd=None, group=1, act=0, acc=None, splits=1):
    from vqa_amd import _lib as L
    if trans:
        K, M = A.shape
        N = B.shape[1]
    else:
        M, K = A.shape
        N = B.shape[0]
    C = acc.clone() if acc is not None else torch.empty(M, N, device='cuda', dtype=out_dtype)
    ws = torch.empty(splits, M, N, device='cuda') if splits > 1 else None
    L.call('vqa_gemm', L.dt(A), int(trans), M, N, K, A, A.stride(0), B, B.stride(0), C, C.stride(0), L.dt(C),
           bias, rowadd, group, act, int(acc is not None), None, 0, ws, splits, None, 0, 0, None)
    return C


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.bfloat16, 2e-6)])
@pytest.mark.parametrize('trans', [0, 1])
@pytest.mark.parametrize('M,N,K', [(64, 64, 32), (256, 3129, 1024), (100, 52, 300 + 4), (3584, 3072, 320),
                                   (17, 1, 64)])
def test_gemm(dtype, tol, trans, M, N, K):
    torch.manual_seed(0)
    # exact-in-bf16 integer-ish data so the bf16 path is compared without input rounding
    A = (torch.randint(-8, 9, (M, K), device='cuda').float() / 8).to(dtype)
    B = (torch.randint(-8, 9, (N, K), device='cuda').float() / 8).to(dtype)
    ref = A.double() @ B.double().t()
    if trans:
        Ain, Bin = A.t().contiguous(), B.t().contiguous()
        if (M % 8) or (N % 8):        # TN vector loads need aligned rows: pad the leading dimension
            Ap = torch.zeros(K, (M + 7) // 8 * 8, device='cuda', dtype=dtype); Ap[:, :M] = Ain; Ain = Ap[:, :M]
            Bp = torch.zeros(K, (N + 7) // 8 * 8, device='cuda', dtype=dtype); Bp[:, :N] = Bin; Bin = Bp[:, :N]
    else:
        Ain, Bin = A, B
    C = _gemm(Ain, Bin, trans)
    err = (C.double() - ref).norm() / ref.norm()
    assert float(err) < tol


def test_gemm_epilogue():
    torch.manual_seed(1)
    M, N, K, G = 50, 72, 64, 25
    A = torch.randn(M, K, device='cuda')
    B = torch.randn(N, K, device='cuda')
    bias = torch.randn(N, device='cuda')
    rowadd = torch.randn(M // G, N, device='cuda')
    acc = torch.randn(M, N, device='cuda')
    ref = torch.relu(A @ B.t() + bias + rowadd.repeat_interleave(G, 0)) + acc
    C = _gemm(A, B, 0, bias=bias, rowadd=rowadd, group=G, act=1, acc=acc)
    assert float((C - ref).norm() / ref.norm()) < 1e-5
    C = _gemm(A, B, 0, bias=bias, rowadd=rowadd, group=G, act=1, acc=acc, splits=3)     # split-K, same epilogue
    assert float((C - ref).norm() / ref.norm()) < 1e-5
    Ct = _gemm(A.t().contiguous(), B.t().contiguous(), 1, splits=2)
    assert float((Ct - A @ B.t()).norm() / (A @ B.t()).norm()) < 1e-5
    Cb = _gemm(A.bfloat16(), B.bfloat16(), 0, out_dtype=torch.bfloat16)
    refb = A.bfloat16().float() @ B.bfloat16().float().t()
    assert float((Cb.float() - refb).norm() / refb.norm()) < 5e-3


def test_probe_tr16():
    """ds_read_b64_tr_b16: lane i of 16-lane group g receives column i of rows 4g..4g+3 (element q = row 4g+q)."""
    from vqa_amd import _lib as L
    tile = torch.arange(256, device='cuda', dtype=torch.float32).reshape(16, 16)   # value = 16*row + col
    out = torch.empty(64, 4, device='cuda')
    L.call('vqa_probe_tr16', tile.bfloat16(), out)
    lane = torch.arange(64, device='cuda')
    g, i = lane // 16, lane % 16
    exp = torch.stack([(4 * g + q) * 16 + i for q in range(4)], dim=1).float()
    assert torch.equal(out, exp), out


@pytest.mark.parametrize('M,N', [(3072, 300), (37, 5), (1, 1), (64, 64), (100, 33)])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_cast_transpose(M, N, dt):
    """vqa_cast_transpose: dst [N][ld] = src [M][lds]^T in the destination dtype, pad columns zero (the GRU's W_ih^T)."""
    from vqa_amd import ops
    torch.manual_seed(M + N)
    src = torch.randn(M, N + 3, device='cuda')[:, :N]                 # a row stride that is not N
    out = ops.cast_transpose(src, dt)
    assert out.shape == (N, M) and out.dtype == dt
    assert torch.equal(out, src.t().to(dt))
    full = out.as_strided((N, ops.pad8(M)), (out.stride(0), 1))
    assert float(full[:, M:].abs().max() if ops.pad8(M) > M else 0.0) == 0.0


@pytest.mark.parametrize('nwords,offset', [(1, 0), (3, 1), (4, 0), (5, 3), (1027, 2), (3_000_000, 0), (65, 1)])
def test_zero_fill(nwords, offset):
    """vqa_zero from any 4-byte aligned address: head words up to the 16-byte boundary, vector body, tail - and nothing
    outside the range."""
    from vqa_amd import _lib as L
    buf = torch.full((nwords + offset + 9,), 7.0, device='cuda')
    L.call('vqa_zero', buf[offset:], nwords * 4)
    torch.cuda.synchronize()
    assert buf[:offset].eq(7).all() and buf[offset:offset + nwords].eq(0).all() and buf[offset + nwords:].eq(7).all()
    with pytest.raises(L.VqaHipError):
        L.call('vqa_zero', buf, 6)                                     # not a multiple of 4 bytes


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,pad,off', [(256, 1024, 0, 0), (3584, 3072, 0, 0), (1500, 304, 8, 0), (256, 301, 3, 0),
                                         (700, 64, 0, 1), (3, 24, 0, 0), (1024, 8, 0, 0), (1025, 520, 16, 0)])
def test_colsum(dtype, M, N, pad, off):
    """vqa_colsum (bias gradients: column sums of a gradient matrix): the 16-byte-load forms (one launch with 16 row
    lanes for M <= 1024, row slabs + combine above), and the scalar form that serves odd widths, odd row strides and
    unaligned views.  Small-integer data: every sum is exact in f32."""
    from vqa_amd import ops
    torch.manual_seed(M + N)
    full = torch.randint(-4, 5, (M, N + pad + off), device='cuda').to(dtype)
    x = full[:, off:off + N]                       # row stride N + pad + off; off = 1: not 16-byte aligned
    got = ops.colsum(x)
    assert torch.equal(got, x.float().sum(0))
    out = torch.full((N,), float('nan'), device='cuda')
    assert ops.colsum(x, out=out) is out and torch.equal(out, got)
    xr = (torch.randn(M, N, device='cuda') * 0.5).to(dtype)
    want = xr.double().sum(0)
    assert float((ops.colsum(xr).double() - want).abs().max()) <= 2e-6 * float(xr.float().abs().sum(0).max()) + 1e-6


@pytest.mark.parametrize('M,N,K,splits', [(256, 1024, 1024, 1), (256, 1024, 3129 + 7, 5), (100, 52, 304, 1), (300, 200, 512, 2)])
def test_gemm_epilogue_multiplier(M, N, K, splits):
    """vqa_gemm's emul / emode against the separate elementwise kernels they replace in the head (vqa_hadamard_fwd,
    vqa_relu_bwd without a mask): the whole-tile epilogue of the small-M path, the ragged-tile fallback and the
    split-K reduce pass - bit for bit."""
    from vqa_amd import _lib as L, ops
    torch.manual_seed(M + K)
    A = torch.randn(M, K, device='cuda').bfloat16()
    B = torch.randn(N, K, device='cuda').bfloat16() * 0.05
    bias = torch.randn(N, device='cuda')
    other = torch.randn(M, N + 8, device='cuda')[:, :N]              # (row-strided multiplier)
    plain = ops.gemm(A, B, bias=bias, act=1, splits=splits)
    # emode 1: the bf16 copy is the product, C itself is untouched
    c2 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    out = ops.gemm(A, B, bias=bias, act=1, splits=splits, c2=c2, emul=other, emode=1)
    want = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    L.call('vqa_hadamard_fwd', L.BF16, other.contiguous(), plain, None, M * N, want, None)
    assert torch.equal(out, plain) and torch.equal(c2, want)
    # emode 16 + act: C scaled by the activation's derivative taken from another tensor (the activation's output)
    for act in (1, 3):
        y = torch.randn(M, N, device='cuda')
        for dt in (torch.bfloat16, torch.float32):
            raw = ops.gemm(A, B, splits=splits)
            got = ops.gemm(A, B, splits=splits, out_dtype=dt, emul=y, emode=16 + act)
            ref = torch.empty(M, N, device='cuda', dtype=dt)
            L.call('vqa_relu_bwd', L.dt(ref), raw, y, None, M * N, ref, act)
            assert torch.equal(got, ref)
    # emode 32 + act: both outputs of vqa_hadamard_bwd from the GEMM that forms its incoming gradient
    for act in (1, 3):
        fv, fq = torch.randn(M, N, device='cuda'), torch.randn(M, N, device='cuda')
        raw = ops.gemm(A, B, splits=splits)
        c2 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        got = ops.gemm(A, B, splits=splits, out_dtype=torch.bfloat16, c2=c2, emul=fv, emul2=fq, emode=32 + act)
        r1, r2 = torch.empty_like(c2), torch.empty_like(c2)
        L.call('vqa_hadamard_bwd', L.BF16, raw, fv, fq, None, M * N, r1, r2, act)
        assert torch.equal(got, r1) and torch.equal(c2, r2)
    with pytest.raises(L.VqaHipError):
        ops.gemm(A, B, emul=other, emode=1)                          # emode 1 needs the bf16 copy
    with pytest.raises(L.VqaHipError):
        ops.gemm(A, B, emul=other, emode=7)
