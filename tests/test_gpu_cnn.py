"""GPU: conv / BN / pool HIP kernels and the whole encoder against PyTorch-CPU fp32 references (the oracle)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {'fp32': torch.float32, 'bf16': torch.bfloat16}


def _ints(shape, lo=-4, hi=5, scale=1.0):
    return torch.randint(lo, hi, shape).float() * scale


def _nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,cin,cout,H,W,pad', [(2, 16, 16, 20, 20, 0), (3, 16, 32, 13, 17, 2), (2, 32, 64, 9, 9, 2),
                                                (1, 64, 64, 30, 30, 0), (2, 128, 256, 14, 14, 0),
                                                (2, 256, 256, 12, 12, 0), (1, 16, 16, 70, 226, 2), (2, 16, 64, 10, 11, 2),
                                                (2, 32, 16, 9, 12, 0), (3, 32, 32, 40, 23, 2),
                                                # two images per tile (ConvArgs::seam): even B, pad 0, multi-chunk, a map
                                                # small enough for a pair; and the same maps with an odd B (unpaired)
                                                (4, 64, 64, 10, 10, 0), (6, 128, 64, 12, 9, 0), (3, 64, 64, 10, 10, 0),
                                                (2, 64, 32, 5, 4, 0)])
def test_conv_fwd_dgrad_wgrad_exact(prec, B, cin, cout, H, W, pad):
    """Small-integer data: every product and sum is exact in bf16/f32, so results must match bit for bit."""
    from vqa_amd import cnn as C
    torch.manual_seed(B * 1000 + cin + cout + H)
    dt = DT[prec]
    x = _ints((B, cin, H, W), -2, 3)
    w = _ints((cout, cin, 3, 3), -2, 3)
    bias = _ints((cout,), -3, 4)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, padding=pad)
    dy = _ints(tuple(y_ref.shape), -2, 3)
    y_ref.backward(dy)

    xd = _nhwc(x, dt)
    wpk = C.pack_weights(w.cuda(), cin, dt, dgrad=False)
    y, stats = C.conv_fwd(xd, wpk, bias.cuda(), cout, pad, True)
    y_st = y_ref.detach().to(dt).float()          # what a bf16 store keeps (sums above 256 round)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), y_st)
    s = stats.sum(0).cpu()
    assert torch.allclose(s[:, 0], y_st.sum((0, 2, 3)), rtol=1e-5)
    assert torch.allclose(s[:, 1], (y_st ** 2).sum((0, 2, 3)), rtol=1e-5)

    dyd = _nhwc(dy, dt)
    wpk_d = C.pack_weights(w.cuda(), cin, dt, dgrad=True)
    dx, _ = C.conv_fwd(dyd, wpk_d, None, cin, 2 - pad, False)
    assert torch.equal(dx.float().cpu().permute(0, 3, 1, 2), xr.grad.to(dt).float())

    dw = C.conv_wgrad(xd, dyd, cin, pad)
    assert torch.equal(dw.cpu(), wr.grad)


@pytest.mark.parametrize('B,cin,cout,H,W,pad', [(1, 64, 128, 19, 21, 0), (2, 128, 64, 33, 18, 2), (3, 256, 256, 12, 12, 0),
                                                (2, 64, 32, 31, 17, 2), (1, 96, 64, 45, 45, 0),
                                                (4, 256, 256, 12, 12, 0), (2, 64, 128, 8, 11, 0)])   # (paired images)
def test_multi_chunk_statistics_are_the_sums_of_the_stored_outputs(B, cin, cout, H, W, pad):
    """bf16 multi-chunk conv (Cin > 32): the BatchNorm statistics are formed by MFMA from the transposed bf16 tile on
    its way out (conv.hip: F^T F and 1^T F over the tile's pixels).  Random data, tiles that hang over the image's
    edges and partly filled fragment slots: the rows must sum to the sums over the STORED outputs (products of bf16
    values are exact in f32; only the f32 summation order differs: rtol 2e-5), and the same call with relu = 1 and no
    statistics must store the same values clamped at 0."""
    from vqa_amd import cnn as C
    torch.manual_seed(cin + cout + H + W)
    dt = torch.bfloat16
    x = torch.randn(B, H, W, cin, device='cuda').to(dt)
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    bias = torch.randn(cout, device='cuda')
    wpk = C.pack_weights(w, cin, dt, dgrad=False)
    y, stats = C.conv_fwd(x, wpk, bias, cout, pad, True)
    y_ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(dt).float(), bias, padding=pad).permute(0, 2, 3, 1)
    assert torch.allclose(y.float(), y_ref, rtol=2e-2, atol=2e-2)
    s = stats.double().sum(0)
    yd = y.double()
    assert torch.allclose(s[:, 0], yd.sum((0, 1, 2)), rtol=2e-5, atol=1e-3)
    assert torch.allclose(s[:, 1], (yd * yd).sum((0, 1, 2)), rtol=2e-5, atol=1e-3)
    y2, none = C.conv_fwd(x, wpk, bias, cout, pad, False, relu=True)
    assert none is None and torch.equal(y2, torch.clamp_min(y, 0))


@pytest.mark.parametrize('img_dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('B,H,W,pad', [(2, 20, 20, 2), (3, 13, 37, 0), (1, 70, 226, 2), (2, 33, 18, 5), (2, 9, 9, 1),
                                       (1, 5, 300, 0), (1, 300, 6, 1)])
def test_first_layer_exact(img_dtype, B, H, W, pad):
    """Dedicated Conv2d(3, 16) kernels on the NCHW image: small-integer data, bit-exact forward, statistics and
    weight gradient (the generic kernels on the channel-padded NHWC copy are the second witness)."""
    from vqa_amd import cnn as C
    torch.manual_seed(B * 100 + H + W + pad)
    x = _ints((B, 3, H, W), -2, 3)
    w = _ints((16, 3, 3, 3), -2, 3)
    bias = _ints((16,), -3, 4)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(x, wr, bias, padding=pad)
    dy = _ints(tuple(y_ref.shape), -2, 3)
    y_ref.backward(dy)
    img = x.to(img_dtype).cuda()
    assert C.first_ok(img, w, torch.bfloat16)
    y, stats = C.first_fwd(img, w.cuda(), bias.cuda(), pad, True)
    y_st = y_ref.detach().bfloat16().float()
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), y_st)
    s = stats.sum(0).cpu()
    assert torch.allclose(s[:, 0], y_st.sum((0, 2, 3)), rtol=1e-5)
    assert torch.allclose(s[:, 1], (y_st ** 2).sum((0, 2, 3)), rtol=1e-5)
    dw = C.first_wgrad(img, _nhwc(dy, torch.bfloat16), pad)
    assert torch.equal(dw.cpu(), wr.grad)
    # second witness: the generic path on the padded NHWC copy
    xd = C.to_nhwc(img, torch.bfloat16)
    y2, _ = C.conv_fwd(xd, C.pack_weights(w.cuda(), C.CIN0_PAD, torch.bfloat16, dgrad=False), bias.cuda(), 16, pad, True)
    assert torch.equal(y2, y)
    # mirrored read (the training loop's whole-batch horizontal flip): equal to the kernels on the flipped batch
    imgf = img.flip(-1).contiguous()
    ym, sm = C.first_fwd(img, w.cuda(), bias.cuda(), pad, True, mirror=True)
    yf, sf = C.first_fwd(imgf, w.cuda(), bias.cuda(), pad, True)
    assert torch.equal(ym, yf) and torch.equal(sm, sf)
    dyd = _nhwc(dy, torch.bfloat16)
    assert torch.equal(C.first_wgrad(img, dyd, pad, mirror=True), C.first_wgrad(imgf, dyd, pad))


@pytest.mark.parametrize('B,H,W,pad', [(2, 20, 20, 2), (3, 13, 37, 0), (1, 70, 226, 2), (2, 33, 18, 5), (1, 5, 300, 0)])
def test_first_layer_fused_bn_wgrad(B, H, W, pad):
    """vqa_conv3x3_first_wgrad_bn (BatchNorm-backward apply inside the weight gradient, dy never stored) against the
    two-kernel path vqa_bn_bwd(apply) -> vqa_conv3x3_first_wgrad on the same inputs, mirrored read included, and both
    against the same gradient in f64 on the CPU.  The fused kernel never rounds a dy to bf16 - it sums x (x) dz, x (x) y
    and x and combines them at the end (dy is affine in dz and y) - so it is the closer of the two to the f64 result."""
    from vqa_amd import _lib as L, cnn as C
    torch.manual_seed(B + H + W + pad)
    OH, OW = H + 2 * pad - 2, W + 2 * pad - 2
    img = torch.rand(B, 3, H, W).half().cuda()
    y = (torch.randn(B, OH, OW, 16) * 1.5 + 0.2).bfloat16().cuda()
    g = torch.randn(B, OH, OW, 16).bfloat16().cuda()
    scale, shift = (torch.rand(16) + 0.5).cuda(), (torch.randn(16) * 0.3).cuda()
    mean, invstd = (torch.randn(16) * 0.2).cuda(), (torch.rand(16) + 0.5).cuda()
    c1, c2 = (torch.randn(16) * 0.05).cuda(), (torch.randn(16) * 0.05).cuda()
    dy = torch.empty_like(y)
    L.call('vqa_bn_bwd', L.dt(y), 1, y, g, dy, scale, shift, mean, invstd, c1, c2, None, B, OH, OW, 16, 0)
    # f64 witness: the same mask (the f32 pre-activation's sign), everything after it in double precision
    yc, gc = y.float().cpu(), g.float().cpu()
    sc, sh, mu, isd, k1, k2 = (t.cpu() for t in (scale, shift, mean, invstd, c1, c2))
    dz = torch.where(torch.addcmul(sh, yc, sc) > 0, gc, torch.zeros(())).double()
    dy64 = sc.double() * (dz - k1.double() - (yc.double() - mu.double()) * isd.double() * k2.double())
    for mirror in (False, True):
        ref = C.first_wgrad(img, dy, pad, mirror=mirror)
        got = C.first_wgrad(img, y, pad, mirror=mirror, bn=(g, scale, shift, mean, invstd, c1, c2))
        assert float((got - ref).norm() / ref.norm()) < 2e-3
        w64 = torch.zeros(16, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        x64 = img.cpu().bfloat16().double()             # (both kernels multiply with the image rounded to bf16)
        F.conv2d(x64.flip(-1) if mirror else x64, w64, padding=pad).backward(dy64.permute(0, 3, 1, 2))
        e_got = float((got.cpu().double() - w64.grad).norm() / w64.grad.norm())
        e_ref = float((ref.cpu().double() - w64.grad).norm() / w64.grad.norm())
        print(f'first_wgrad_bn vs f64: fused {e_got:.2e}, two kernels {e_ref:.2e}')
        assert e_got < 1e-6 and e_got <= e_ref, (e_got, e_ref)
        # per batch slice (the sums need only scale and shift), slabs combined at the end: the same gradient
        parts = []
        for b in range(B):
            parts.append(torch.empty(C.first_wgrad_rows(1, OH, OW), 3, 16, 32, device='cuda'))
            assert C.first_wgrad(img[b:b + 1], y[b:b + 1], pad, mirror=mirror, bn=(g[b:b + 1], scale, shift),
                                 part_out=parts[-1]) is None
        sl = C.first_wgrad_reduce(torch.cat(parts), scale, mean, invstd, c1, c2)
        assert float((sl.cpu().double() - w64.grad).norm() / w64.grad.norm()) < 1e-6


@pytest.mark.parametrize('prec,tol', [('fp32', 1e-5), ('bf16', 2e-2)])
@pytest.mark.parametrize('pool', [0, 1, 2])
@pytest.mark.parametrize('C', [16, 64, 256])
@pytest.mark.parametrize('H,W', [(10, 12), (11, 13), (2, 3)])
def test_bn_relu_pool_fwd_bwd(prec, tol, pool, C, H, W):
    from vqa_amd import _lib as L, cnn as M
    from oracle import vqa_oracle as O
    import ctypes
    torch.manual_seed(C + pool)
    dt = DT[prec]
    B = 3
    y = (torch.randn(B, C, H, W) * 1.5 + 0.3).to(dt).float()
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.3
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z, mean, var = O._BatchNormTrain.apply(yr, gr, br)
    a_ref = torch.relu(z)
    if pool == 1:
        a_ref = F.max_pool2d(a_ref, 2, 2)
    elif pool == 2:
        a_ref = F.avg_pool2d(a_ref, 2, 2)
    da = torch.randn_like(a_ref).to(dt).float()
    a_ref.backward(da)

    yd = _nhwc(y, dt)
    # statistics the way the conv epilogue hands them over: one row of (sum, sumsq)
    part = torch.stack([y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))], dim=1).reshape(1, C, 2).cuda()
    f = lambda: torch.empty(C, device='cuda')
    mean_d, invstd, scale, shift = f(), f(), f(), f()
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    n = float(B * H * W)
    gd, bd = gamma.cuda(), beta.cuda()      # keep the device copies alive across the asynchronous launch
    nbt = torch.full((), 6, device='cuda', dtype=torch.int64)
    L.call('vqa_bn_finalize', part, 1, C, n, gd, bd, rm, rv, 0.1, 1e-5, mean_d, invstd, scale, shift, nbt)
    assert int(nbt) == 7                                    # num_batches_tracked is bumped by the statistics kernel
    assert torch.allclose(mean_d.cpu(), mean, atol=1e-5)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * var * n / (n - 1), rtol=1e-4)
    a = M.bn_act_pool(yd, scale, shift, pool)
    err = (a.float().cpu().permute(0, 3, 1, 2) - a_ref.detach()).norm() / a_ref.detach().norm()
    assert float(err) < (1e-6 if prec == 'fp32' else 4e-3)

    dad = _nhwc(da, dt)
    rows = L.lib().vqa_bn_bwd_rows(L.dt(yd), B, H, W, C, pool)
    partb = torch.empty(rows, C, 2, device='cuda')
    dgamma, dbeta, c1, c2 = f(), f(), f(), f()
    dy = torch.empty_like(yd)
    L.call('vqa_bn_bwd', L.dt(yd), 0, yd, dad, None, scale, shift, mean_d, invstd, None, None, partb, B, H, W, C, pool)
    L.check(L.lib().vqa_bn_bwd_finalize(L.stream(), L.ptr(partb), rows, C, ctypes.c_double(n), L.ptr(dgamma),
                                        L.ptr(dbeta), L.ptr(c1), L.ptr(c2)), 'bn_bwd_finalize')
    L.call('vqa_bn_bwd', L.dt(yd), 1, yd, dad, dy, scale, shift, mean_d, invstd, c1, c2, None, B, H, W, C, pool)
    rel = lambda u, v: float((u - v).norm() / v.norm())
    assert rel(dgamma.cpu(), gr.grad) < tol
    assert rel(dbeta.cpu(), br.grad) < tol
    assert rel(dy.float().cpu().permute(0, 3, 1, 2), yr.grad) < tol


@pytest.mark.parametrize('prec,tol', [('fp32', 2e-5), ('bf16', 6e-3)])
@pytest.mark.parametrize('cin,cout,S,pad,pool', [(16, 16, 40, 0, 1), (16, 32, 21, 2, 0), (64, 64, 30, 0, 2),
                                                 (128, 256, 14, 0, 0), (256, 256, 12, 0, 1)])
def test_conv_block_fwd_bwd(prec, tol, cin, cout, S, pad, pool):
    """One conv+BN+ReLU(+pool) block from IDENTICAL inputs, real-valued data: output and every gradient against
    the oracle's formulas (for bf16: with the same storage rounding points, so the comparison is tight)."""
    from vqa_amd import cnn as M
    from oracle import vqa_oracle as O
    torch.manual_seed(cin + cout + S)
    dt = DT[prec]
    emu = torch.bfloat16 if prec == 'bf16' else None
    B = 3
    x = torch.relu(torch.randn(B, cin, S, S)).to(dt).float()
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    b, gamma, beta = torch.randn(cout) * 0.1, torch.rand(cout) + 0.5, torch.randn(cout) * 0.3
    xr, wr, gr, br = [t.clone().requires_grad_(True) for t in (x, w, gamma, beta)]
    y = O._q(F.conv2d(xr, O._q(wr, emu), b, padding=pad), emu, True, True)
    z, _, _ = O._BatchNormTrain.apply(y, gr, br)
    a_ref = torch.relu(z)
    a_ref = F.max_pool2d(a_ref, 2, 2) if pool == 1 else (F.avg_pool2d(a_ref, 2, 2) if pool == 2 else a_ref)
    a_ref = O._q(a_ref, emu, True, False)
    da = torch.randn_like(a_ref).to(dt).float()
    a_ref.backward(da)

    dev = lambda t: t.cuda()
    rm, rv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
    a, saved = M.block_fwd(_nhwc(x, dt), dev(w), dev(b), dev(gamma), dev(beta), rm, rv, pad, pool, True)
    dw, dg, dbt = torch.empty(cout, cin, 3, 3, device='cuda'), torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
    dx, _ = M.block_bwd(_nhwc(da, dt), saved, dev(w), cin, pad, pool, True, (dw, dg, dbt))
    rel = lambda u, v: float((u.double().cpu() - v.double()).norm() / v.double().norm())
    assert rel(a.float().permute(0, 3, 1, 2), a_ref.detach()) < tol
    assert rel(dg, gr.grad) < tol and rel(dbt, br.grad) < tol
    assert rel(dw, wr.grad) < tol
    # dx: the oracle's gradient is not rounded on arrival at a leaf; bf16 storage adds 2^-9 per element
    assert rel(dx.float().permute(0, 3, 1, 2), xr.grad) < (tol if prec == 'fp32' else 8e-3)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('padding,pooling,size,B', [(2, 'max', 224, 2), (0, 'avg', 224, 2), (5, 'max', 224, 1),
                                                   (2, 'max', 448, 1)])
def test_cnn_stack_vs_oracle(prec, padding, pooling, size, B):
    """Whole encoder, train and eval mode, forward + parameter gradients + running statistics vs the oracle."""
    from vqa_amd import cnn as M, synth
    from oracle import vqa_oracle as O
    sd = synth.state_dict(42, 64, 50, 30)
    csd = {k[4:]: v for k, v in sd.items() if k.startswith('cnn.')}
    img = synth.images(7, B, size)
    dv = torch.rand(B, {(2, 224): 25, (0, 224): 9, (5, 224): 49, (2, 448): 144}[(padding, size)], 256)
    # oracle
    p = {k: (v.clone().requires_grad_(True) if O.is_param(k) else v) for k, v in sd.items() if k.startswith('cnn.')}
    new_stats = {}
    emu = torch.bfloat16 if prec == 'bf16' else None   # bf16 kernels are checked against the same rounding points
    v_ref = O.cnn_forward(p, img.float(), padding, pooling, True, new_stats, act_dtype=emu)
    names = [k for k in p if O.is_param(k)]
    g_ref = dict(zip(names, torch.autograd.grad((v_ref * dv).sum(), [p[k] for k in names])))
    v_eval = O.cnn_forward(sd, img.float(), padding, pooling, False, act_dtype=emu)
    # HIP
    m = M.CNN(padding, pooling, precision=prec)
    m.load_state_dict(csd)
    m = m.cuda()
    m.eval()
    with torch.no_grad():
        out_eval = m(img.cuda())
    m.train()
    out = m(img.cuda())
    (out * dv.cuda()).sum().backward()
    rel = lambda u, v: float((u.detach().cpu().double() - v.detach().double()).norm() / (v.detach().double().norm() + 1e-30))
    tol_f, tol_g = (1e-4, 3e-2) if prec == 'fp32' else (5e-2, 5e-1)
    assert rel(out_eval, v_eval) < tol_f
    assert rel(out, v_ref) < tol_f
    for k, buf in m.state_dict().items():
        if 'running' in k:
            assert rel(buf, new_stats['cnn.' + k]) < tol_f, k
    worst = 0.0
    for k, prm in m.named_parameters():
        ref = g_ref['cnn.' + k]
        if k.endswith('.bias') and int(k.split('.')[1]) in O.CONV_IDX:
            assert float(prm.grad.abs().max()) == 0.0
            continue
        e = rel(prm.grad, ref)
        worst = max(worst, e)
        assert e < tol_g, (k, e)
    print(flush=True); print(f'[{prec} pad{padding} {pooling} {size}] fwd {rel(out, v_ref):.2e} eval {rel(out_eval, v_eval):.2e} '
          f'worst grad {worst:.2e}')


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 16, 16, 40, 40), (3, 32, 32, 23, 31), (2, 64, 64, 30, 30),
                                            (2, 128, 128, 16, 14), (3, 256, 256, 12, 12), (1, 16, 16, 226, 226),
                                            (2, 16, 32, 9, 20), (1, 64, 128, 17, 17)])
def test_fused_input_bn_relu_is_the_separate_pass(prec, B, cin, cout, H, W):
    """vqa_conv3x3_fwd / vqa_conv3x3_wgrad with in_scale / in_shift (the previous layer's BatchNorm + ReLU applied while
    the input tile is staged) against the separate pass followed by the plain kernels: y, the BatchNorm statistics and
    dW bit for bit - same arithmetic, the activation just never goes through HBM; and against the fp32 formula."""
    from vqa_amd import cnn as C
    torch.manual_seed(cin * 7 + cout + H)
    dt = DT[prec]
    yprev = (torch.randn(B, H, W, cin) * 1.5).to(dt).cuda()                 # raw output of the previous conv, NHWC
    scale = (torch.rand(cin) + 0.5).cuda()
    shift = (torch.randn(cin) * 0.4).cuda()
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    bias = torch.randn(cout) * 0.1
    wpk = C.pack_weights(w.cuda(), cin, dt, dgrad=False)
    a = C.bn_act_pool(yprev, scale, shift, 0)
    y_sep, st_sep = C.conv_fwd(a, wpk, bias.cuda(), cout, 0, True)
    y_fus, st_fus = C.conv_fwd(yprev, wpk, bias.cuda(), cout, 0, True, in_coeffs=(scale, shift))
    assert torch.equal(y_fus, y_sep) and torch.equal(st_fus, st_sep)
    a_ref = torch.relu(yprev.float().cpu() * scale.cpu() + shift.cpu()).to(dt).float().permute(0, 3, 1, 2)
    y_ref = F.conv2d(a_ref, w.to(dt).float(), bias)
    err = (y_fus.float().cpu().permute(0, 3, 1, 2) - y_ref).norm() / y_ref.norm()
    assert float(err) < (1e-5 if prec == 'fp32' else 4e-3)
    dy = (torch.randn(B, H - 2, W - 2, cout)).to(dt).cuda()
    dw_sep = C.conv_wgrad(a, dy, cin, 0)
    dw_fus = C.conv_wgrad(yprev, dy, cin, 0, in_coeffs=(scale, shift))
    assert torch.equal(dw_fus, dw_sep)
    # a padded conv cannot take the fused form (its zero padding would become relu(shift)): refused, not wrong
    from vqa_amd._lib import VqaHipError
    with pytest.raises(VqaHipError):
        C.conv_fwd(yprev, wpk, bias.cuda(), cout, 2, True, in_coeffs=(scale, shift))


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('padding,pooling,size,B', [(2, 'max', 224, 3), (0, 'avg', 224, 2), (5, 'max', 200, 2)])
def test_encoder_with_fused_bn_forward_is_bit_identical(prec, padding, pooling, size, B, monkeypatch):
    """The whole encoder, training mode: BatchNorm + ReLU of each block's first conv applied by its consumers
    (cnn.FUSE_BN_FWD, the default) against the separate passes: output, running statistics and every gradient bit
    for bit."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(9, B, size).cuda()
    # (vqa_conv3x3_dgrad_wgrad exists only for the fused-forward form and sums its weight-gradient slabs over another
    # grid: off here, so that both runs issue the same weight-gradient launches; it has its own tests below)
    monkeypatch.setattr(M, 'FUSE_DGRAD_WGRAD', 0)
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(M, 'FUSE_BN_FWD', fuse)
        m = M.CNN(padding, pooling, precision=prec)
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = m(img)
        torch.manual_seed(1)
        (out * torch.rand_like(out)).sum().backward()
        outs.append((out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()},
                     {k: b.clone() for k, b in m.named_buffers()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,c_lo,c_hi,H,W', [(2, 16, 16, 40, 40), (3, 32, 32, 23, 31), (2, 64, 64, 30, 30),
                                             (2, 128, 128, 16, 14), (3, 256, 256, 12, 12), (1, 16, 16, 226, 226),
                                             (2, 16, 32, 9, 20)])
def test_dgrad_with_fused_bn_backward_reduction(prec, B, c_lo, c_hi, H, W):
    """vqa_conv3x3_fwd as a data gradient with bn_y / bn_coef: the output d is bit-identical to the plain call, and the
    rows it returns, finalized, equal the BatchNorm-backward sums of the separate reduce pass over (y, d)
    (vqa_bn_bwd apply = 0, pool = 0) up to fp32 summation order - and the fp64 formula."""
    import ctypes
    from vqa_amd import _lib as L, cnn as C
    torch.manual_seed(c_lo * 3 + c_hi + H)
    dt = DT[prec]
    # upper conv: c_lo -> c_hi channels, pad 0, input [B,H,W,c_lo] (the lower block's activation), output H-2 x W-2
    w = torch.randn(c_hi, c_lo, 3, 3) / (3 * c_lo ** 0.5)
    dy = torch.randn(B, H - 2, W - 2, c_hi).to(dt).cuda()
    y_lo = (torch.randn(B, H, W, c_lo) * 1.5).to(dt).cuda()                  # lower layer's raw conv output
    coef = torch.stack([torch.rand(c_lo) + 0.5, torch.randn(c_lo) * 0.4, torch.randn(c_lo) * 0.3,
                        torch.rand(c_lo) + 0.7]).cuda()                       # scale, shift, mean, invstd
    wd = C.pack_weights(w.cuda(), c_lo, dt, dgrad=True)
    d_plain, _ = C.conv_fwd(dy, wd, None, c_lo, 2, False, kind='dgrad')
    d_fused, rows = C.conv_fwd(dy, wd, None, c_lo, 2, False, kind='dgrad', bst=(y_lo, coef))
    assert torch.equal(d_fused, d_plain)
    n = float(B * H * W)
    f = lambda: torch.empty(c_lo, device='cuda')
    out = []
    nrows = L.lib().vqa_bn_bwd_rows(L.dt(y_lo), B, H, W, c_lo, 0)
    part = torch.empty(nrows, c_lo, 2, device='cuda')
    L.call('vqa_bn_bwd', L.dt(y_lo), 0, y_lo, d_plain, None, coef[0], coef[1], coef[2], coef[3], None, None, part,
           B, H, W, c_lo, 0)
    for p in (rows, part):
        dgamma, dbeta, c1, c2 = f(), f(), f(), f()
        L.check(L.lib().vqa_bn_bwd_finalize(L.stream(), L.ptr(p), p.shape[0], c_lo, ctypes.c_double(n), L.ptr(dgamma),
                                            L.ptr(dbeta), L.ptr(c1), L.ptr(c2)), 'bn_bwd_finalize')
        out.append((dgamma.cpu().double(), dbeta.cpu().double()))
    yl, dd, cf = y_lo.double().cpu(), d_plain.double().cpu(), coef.double().cpu()
    dz = torch.where(yl * cf[0] + cf[1] > 0, dd, torch.zeros_like(dd))
    want = ((dz * (yl - cf[2]) * cf[3]).sum((0, 1, 2)), dz.sum((0, 1, 2)))
    scale = [float(t.abs().max()) for t in want]
    for got in out:
        for k in range(2):
            assert float((got[k] - want[k]).abs().max()) < 2e-5 * max(scale[k], 1.0) * (n ** 0.5)
    for k in range(2):
        assert float((out[0][k] - out[1][k]).abs().max()) < 1e-5 * max(scale[k], 1.0) * (n ** 0.5)


@pytest.mark.parametrize('B,c,H,W', [(2, 16, 40, 40), (3, 32, 23, 31), (1, 16, 226, 226), (2, 32, 116, 116),
                                      (5, 16, 9, 20), (1, 32, 3, 3), (64, 16, 37, 50)])
def test_dgrad_and_wgrad_from_one_kernel(B, c, H, W):
    """vqa_conv3x3_dgrad_wgrad against the two kernels it replaces (the BST data gradient and the XF weight gradient):
    dx and the BatchNorm-backward rows bit-identical; the weight gradient equal up to the fp32 summation order of the
    partial slabs, and exact on integer-valued data (every product and sum exact in f32)."""
    from vqa_amd import cnn as C
    dt = torch.bfloat16
    for exact in (True, False):
        torch.manual_seed(c + H + exact)
        w = torch.randn(c, c, 3, 3) / (3 * c ** 0.5)
        if exact:
            dy = torch.randint(-3, 4, (B, H - 2, W - 2, c)).to(dt).cuda()
            y_lo = torch.randint(-4, 5, (B, H, W, c)).to(dt).cuda()
            coef = torch.stack([torch.randint(1, 3, (c,)).float(), torch.randint(-2, 3, (c,)).float(),
                                torch.zeros(c), torch.ones(c)]).cuda()
        else:
            dy = torch.randn(B, H - 2, W - 2, c).to(dt).cuda()
            y_lo = (torch.randn(B, H, W, c) * 1.5).to(dt).cuda()
            coef = torch.stack([torch.rand(c) + 0.5, torch.randn(c) * 0.4, torch.randn(c) * 0.3,
                                torch.rand(c) + 0.7]).cuda()
        wd = C.pack_weights(w.cuda(), c, dt, dgrad=True)
        d_ref, rows_ref = C.conv_fwd(dy, wd, None, c, 2, False, kind='dgrad', bst=(y_lo, coef))
        dw_ref = C.conv_wgrad(y_lo, dy, c, 0, in_coeffs=(coef[0], coef[1]))
        dx, rows, part = C.conv_dgrad_wgrad(dy, wd, c, (y_lo, coef), c)
        dw = torch.full((c, c, 3, 3), float('nan'), device='cuda')
        C.wgrad_reduce_all([(part, dw, part.shape[0], c, c, c)])
        assert torch.equal(dx, d_ref)
        # (the fused form has its own persistent grid: compare the finalized sums, not the rows)
        assert rows.shape[1:] == rows_ref.shape[1:]
        s, sr = rows.double().sum(0), rows_ref.double().sum(0)
        assert float((s - sr).abs().max()) <= 1e-5 * max(float(sr.abs().max()), 1.0) * (B * H * W) ** 0.5
        if exact:
            assert torch.equal(dw, dw_ref)
            xa = torch.relu(y_lo.float() * coef[0] + coef[1]).permute(0, 3, 1, 2).double().cpu()
            g = dy.float().permute(0, 3, 1, 2).double().cpu()
            want = torch.nn.grad.conv2d_weight(xa, (c, c, 3, 3), g)
            assert torch.equal(dw.double().cpu(), want)
        else:
            tol = 2e-6 * float(dw_ref.abs().max()) * (B * H * W) ** 0.5
            assert float((dw - dw_ref).abs().max()) <= tol


@pytest.mark.parametrize('streams', [True, False])
def test_encoder_with_one_kernel_gradients_matches_two(streams, monkeypatch):
    """Whole encoder, bf16: cnn.FUSE_DGRAD_WGRAD = 32 (blocks 1 and 2, block 1 in batch slices) vs 0."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(3, 8, 224).cuda()              # (seed 3, a batch of 8)
    monkeypatch.setattr(M, 'WGRAD_STREAM', streams)
    monkeypatch.setattr(M, 'BWD_SLICES', 2)
    monkeypatch.setattr(M, 'BWD_SLICE_MIN_MB', 1.0)
    outs = []
    for fuse in (32, 0):
        monkeypatch.setattr(M, 'FUSE_DGRAD_WGRAD', fuse)
        m = M.CNN(2, 'max', precision='bf16')
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = m(img)
        torch.manual_seed(1)
        (out * torch.rand_like(out)).sum().backward()
        torch.cuda.synchronize()
        outs.append((out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    # layers 5-10 (above both fused convs) see the same launches: bit-identical.  Below, the BatchNorm-backward sums of
    # layers 1 and 3 come from another persistent grid (fp32 summation order) and dy is re-rounded to bf16 from there on.
    same = 0
    for k in outs[0][1]:
        a, b = outs[0][1][k].double(), outs[1][1][k].double()
        same += int(torch.equal(a, b))
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0, k
        else:
            assert float((a - b).norm() / b.norm()) < 1e-2, k
    assert same >= 24


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_encoder_with_fused_bn_backward_matches_separate_pass(prec, monkeypatch):
    """Whole encoder, cnn.FUSE_BN_BWD on (default) vs off: forward identical; gradients equal up to the fp32 summation
    order of the BatchNorm-backward sums (1e-5 relative per tensor: the sums feed every gradient below them)."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(9, 3, 224).cuda()
    outs = []
    for fuse in (32, 0):
        monkeypatch.setattr(M, 'FUSE_BN_BWD', fuse)
        m = M.CNN(2, 'max', precision=prec)
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = m(img)
        torch.manual_seed(1)
        (out * torch.rand_like(out)).sum().backward()
        outs.append((out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        a, b = outs[0][1][k].double(), outs[1][1][k].double()
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0
        else:
            assert float((a - b).norm() / b.norm()) < (1e-5 if prec == 'fp32' else 1e-2), k   # (bf16: dy re-rounded)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,c_lo,c_hi,H,W', [(2, 16, 32, 40, 40), (3, 32, 64, 22, 30), (1, 16, 32, 224, 224),
                                             (2, 64, 128, 12, 16)])
def test_pooled_bn_backward_sums_from_ysel(prec, B, c_lo, c_hi, H, W):
    """Max-pooled block below a conv (cnn.py:27-30): the pool pass's second output ysel is the raw y at the position
    the pool selects (first maximum of z in window order, negative scales included), its `a` is unchanged, and the data
    gradient above with bn_y = ysel returns rows that, finalized, equal the pooled reduce pass over the full-resolution
    (y, d) (vqa_bn_bwd apply = 0, pool = 1) up to fp32 summation order."""
    import ctypes
    from vqa_amd import _lib as L, cnn as C
    torch.manual_seed(c_lo + c_hi + H)
    dt = DT[prec]
    y = (torch.randn(B, H, W, c_lo) * 1.5).to(dt)
    y[:, 0:2, 0:2] = y[:, 0:1, 0:1]                                          # a window of ties
    y = y.cuda()
    coef = torch.stack([(torch.rand(c_lo) + 0.5) * torch.where(torch.arange(c_lo) % 5 == 0, -1.0, 1.0),
                        torch.randn(c_lo) * 0.4, torch.randn(c_lo) * 0.3, torch.rand(c_lo) + 0.7]).cuda()
    a_plain = C.bn_act_pool(y, coef[0], coef[1], 1)
    a, ysel = C.bn_act_pool(y, coef[0], coef[1], 1, want_ysel=True)
    assert ysel is not None and torch.equal(a, a_plain)
    # definition: windows in (0,0), (0,1), (1,0), (1,1) order, first maximum of z
    yw = torch.stack([y[:, i::2, j::2][:, :H // 2, :W // 2] for i in (0, 1) for j in (0, 1)], 0).float()
    z = torch.addcmul(coef[1], yw, coef[0])                                  # fma like the kernel
    first = z.argmax(0, keepdim=True)                                        # torch: first maximum
    assert torch.equal(ysel.float(), yw.gather(0, first)[0])
    # upper conv: c_lo -> c_hi, padding 2 (the first conv of the next block), on the pooled map
    PH, PW = H // 2, W // 2
    w = torch.randn(c_hi, c_lo, 3, 3) / (3 * c_lo ** 0.5)
    dy = torch.randn(B, PH + 2, PW + 2, c_hi).to(dt).cuda()
    wd = C.pack_weights(w.cuda(), c_lo, dt, dgrad=True)
    d_plain, _ = C.conv_fwd(dy, wd, None, c_lo, 0, False, kind='dgrad')
    d_fused, rows = C.conv_fwd(dy, wd, None, c_lo, 0, False, kind='dgrad', bst=(ysel, coef))
    assert torch.equal(d_fused, d_plain) and tuple(d_plain.shape) == (B, PH, PW, c_lo)
    nrows = L.lib().vqa_bn_bwd_rows(L.dt(y), B, H, W, c_lo, 1)
    part = torch.empty(nrows, c_lo, 2, device='cuda')
    L.call('vqa_bn_bwd', L.dt(y), 0, y, d_plain, None, coef[0], coef[1], coef[2], coef[3], None, None, part,
           B, H, W, c_lo, 1)
    n = float(B * H * W)
    out = []
    for p in (rows, part):
        dgamma, dbeta, c1, c2 = (torch.empty(c_lo, device='cuda') for _ in range(4))
        L.check(L.lib().vqa_bn_bwd_finalize(L.stream(), L.ptr(p), p.shape[0], c_lo, ctypes.c_double(n), L.ptr(dgamma),
                                            L.ptr(dbeta), L.ptr(c1), L.ptr(c2)), 'bn_bwd_finalize')
        out.append((dgamma.cpu().double(), dbeta.cpu().double()))
    scale = [max(float(t.abs().max()), 1.0) for t in out[1]]
    for k in range(2):
        assert float((out[0][k] - out[1][k]).abs().max()) < 1e-5 * scale[k] * (n ** 0.5)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_encoder_with_pooled_bn_backward_matches_separate_pass(prec, monkeypatch):
    """Whole encoder, cnn.FUSE_POOL_BWD off / 16 (the 224x224 block only) / 32 (also through the multi-chunk data
    gradient) / 128 (every pooled block with a conv above it: the default): forward identical; gradients equal up to the
    fp32 summation order of the sums."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(9, 3, 224).cuda()
    outs = []
    for fuse in (0, 16, 32, 128):
        monkeypatch.setattr(M, 'FUSE_POOL_BWD', fuse)
        m = M.CNN(2, 'max', precision=prec)
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = m(img)
        torch.manual_seed(1)
        (out * torch.rand_like(out)).sum().backward()
        outs.append((out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0])
        for k in o[1]:
            a, b = o[1][k].double(), outs[0][1][k].double()
            if float(b.norm()) == 0.0:
                assert float(a.norm()) == 0.0
            else:
                assert float((a - b).norm() / b.norm()) < (1e-5 if prec == 'fp32' else 1e-2), k


@pytest.mark.parametrize('prec,first_sliced', [('fp32', False), ('bf16', False), ('bf16', True)])
@pytest.mark.parametrize('streams', [True, False])
def test_batch_sliced_backward_matches_whole_batch(prec, first_sliced, streams, monkeypatch):
    """cnn.BWD_SLICES: the backward of a block in batch slices (apply pass -> weight gradient -> data gradient per
    slice; by default only where a tensor exceeds the Infinity Cache) against the whole-batch form, forced here for
    EVERY block (2 and 4 slices of a batch of 8): data gradients and BatchNorm gradients are the same arithmetic per
    image, weight gradients and BatchNorm-backward sums differ in fp32 summation order only.  first_sliced
    (cnn.FIRST_WGRAD_SLICED): the first layer's weight gradient too is formed per slice of block 1 - its three sums
    right behind each slice's data gradient, combined once BatchNorm 1's c1 and c2 exist."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(3, 8, 224).cuda()              # (seed 3, a batch of 8)
    monkeypatch.setattr(M, 'WGRAD_STREAM', streams)
    monkeypatch.setattr(M, 'BWD_SLICE_MIN_MB', 0.0)
    monkeypatch.setattr(M, 'FIRST_WGRAD_SLICED', first_sliced)
    combined, reduce0 = [], M.first_wgrad_reduce
    monkeypatch.setattr(M, 'first_wgrad_reduce', lambda part, *a, **k: (combined.append(part.shape[0]), reduce0(part, *a, **k))[1])
    outs = []
    for slices in (1, 2, 4):
        monkeypatch.setattr(M, 'BWD_SLICES', slices)
        m = M.CNN(2, 'max', precision=prec)
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = m(img)
        torch.manual_seed(1)
        (out * torch.rand_like(out)).sum().backward()
        torch.cuda.synchronize()
        outs.append((out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert M._bwd_slices(8, 1e9) == 4 and M._bwd_slices(6, 1e9) == 2
    assert len(combined) == (2 if first_sliced else 0)          # (the 2- and the 4-slice pass)
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0])
        for k in o[1]:
            a, b = o[1][k].double(), outs[0][1][k].double()
            if float(b.norm()) == 0.0:
                assert float(a.norm()) == 0.0
            else:
                assert float((a - b).norm() / b.norm()) < (1e-5 if prec == 'fp32' else 1e-2), k


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('streams', [True, False])
def test_deferred_weight_gradient_reduction_is_bit_identical(prec, streams, monkeypatch):
    """cnn.DEFER_WGRAD_REDUCE (opt-in): the nine weight gradients leave their partial slabs and ONE launch
    (vqa_conv3x3_wgrad_reduce_all) sums them at the end of the backward - the same sums in the same order as a reduce
    per layer, with and without the weight-gradient stream."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(9, 3, 224).cuda()
    monkeypatch.setattr(M, 'WGRAD_STREAM', streams)
    outs = []
    for defer in (True, False):
        monkeypatch.setattr(M, 'DEFER_WGRAD_REDUCE', defer)
        m = M.CNN(2, 'max', precision=prec)
        m.load_state_dict(sd)
        m = m.cuda().train()
        out = m(img)
        torch.manual_seed(1)
        (out * torch.rand_like(out)).sum().backward()
        torch.cuda.synchronize()
        outs.append({k: p.grad.clone() for k, p in m.named_parameters()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('cin,cout,H,W', [(256, 256, 12, 12), (64, 128, 10, 7), (128, 64, 6, 14)])
def test_paired_and_unpaired_tiles_store_the_same_bits(prec, cin, cout, H, W):
    """Two images per tile (conv.hip, ConvArgs::seam) against one image per tile on RANDOM data: an even batch is paired,
    the same images in an odd batch are not - outputs bit for bit (training form with statistics, eval form with the
    ReLU), statistics equal up to the fp32 order of their partial rows (one row per pair instead of per image)."""
    from vqa_amd import cnn as C
    dt = DT[prec]
    torch.manual_seed(cin + H)
    x = torch.randn(6, H, W, cin, device='cuda').to(dt)
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    bias = torch.randn(cout, device='cuda')
    wpk = C.pack_weights(w, cin, dt, dgrad=False)
    y6, s6 = C.conv_fwd(x, wpk, bias, cout, 0, True)
    y5, s5 = C.conv_fwd(x[:5].contiguous(), wpk, bias, cout, 0, True)
    assert s6.shape[0] == 3 and s5.shape[0] == 5           # (pairs / images: the forms really differ)
    assert torch.equal(y6[:5], y5)
    y1, s1 = C.conv_fwd(x[5:].contiguous(), wpk, bias, cout, 0, True)
    tot = s5.double().sum(0) + s1.double().sum(0)
    assert torch.allclose(s6.double().sum(0), tot, rtol=1e-5, atol=1e-3)
    r6, _ = C.conv_fwd(x, wpk, bias, cout, 0, False, relu=True)
    r5, _ = C.conv_fwd(x[:5].contiguous(), wpk, bias, cout, 0, False, relu=True)
    assert torch.equal(r6[:5], r5) and torch.equal(r6, torch.clamp_min(y6, 0))
    # the weight gradient pairs the same way (bf16; vqa_conv3x3_wgrad's seam): a sum over the batch, so the paired call
    # on six images must equal the unpaired calls on five + one up to the fp32 order of the pixel reduction - plain and
    # with the previous layer's BatchNorm + ReLU applied to x on the way in
    dy = torch.randn(6, H - 2, W - 2, cout, device='cuda').to(dt)
    coef = (torch.rand(cin, device='cuda') + 0.5, torch.randn(cin, device='cuda') * 0.3)
    for ic in (None, coef):
        if prec == 'bf16':
            assert C._wgrad_geom(x, dy, 0)[3] == H - 2 and C._wgrad_geom(x[:5], dy[:5], 0)[3] == 0
        d6 = C.conv_wgrad(x, dy, cin, 0, in_coeffs=ic)
        d51 = (C.conv_wgrad(x[:5].contiguous(), dy[:5].contiguous(), cin, 0, in_coeffs=ic) +
               C.conv_wgrad(x[5:].contiguous(), dy[5:].contiguous(), cin, 0, in_coeffs=ic))
        assert float((d6 - d51).abs().max()) <= 2e-6 * float(d51.abs().max()) * (6 * H * W) ** 0.5


@pytest.mark.parametrize('B,H,W,cin,cout', [(2, 34, 50, 16, 16), (3, 18, 18, 32, 32), (1, 66, 34, 16, 32), (2, 226, 226, 16, 16),
                                            (2, 58, 58, 64, 64), (3, 30, 30, 128, 128), (1, 30, 38, 64, 128), (2, 22, 38, 256, 64), (1, 14, 62, 96, 32)])
def test_conv_relu_maxpool_epilogue(B, H, W, cin, cout):
    """vqa_conv3x3_fwd with relu = 3 (inference: conv + folded BatchNorm + ReLU + the 2x2 max pool from one kernel, the
    full-resolution activation never written) against conv -> ReLU -> max_pool2d on the CPU (small integers: exact) and,
    on random data, bit for bit against the two-kernel path relu = 1 -> vqa_bn_act_pool_fwd."""
    from vqa_amd import cnn as C
    torch.manual_seed(B * H + W + cin)
    for exact in (True, False):
        if exact:
            x, w, bias = _ints((B, cin, H, W), -2, 3), _ints((cout, cin, 3, 3), -2, 3), _ints((cout,), -3, 4)
        else:
            x, w, bias = torch.randn(B, cin, H, W), torch.randn(cout, cin, 3, 3) * 0.1, torch.randn(cout) * 0.1
        xd = _nhwc(x, torch.bfloat16)
        assert C.conv_pool_ok(xd, 0)
        wp = C.pack_weights(w.cuda(), cin, torch.bfloat16, dgrad=False)
        got, _ = C.conv_fwd(xd, wp, bias.cuda(), cout, 0, False, relu=True, pool=True)
        y, _ = C.conv_fwd(xd, wp, bias.cuda(), cout, 0, False, relu=True)
        one, zero = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
        two = C.bn_act_pool(y, one, zero, 1)
        assert got.shape == two.shape and torch.equal(got, two)
        if exact:
            ref = F.max_pool2d(F.relu(F.conv2d(x, w, bias)), 2, 2)
            assert torch.equal(got.float().cpu().permute(0, 3, 1, 2), ref.bfloat16().float())


def test_eval_forward_with_pooled_epilogue_is_bit_identical(monkeypatch):
    """cnn.EVAL_FUSE_POOL (default): the inference pass of the encoder with blocks 1 and 2 pooling in their second
    conv's epilogue against the same pass with the pool-only kernels."""
    from vqa_amd import cnn as M, synth
    sd = {k[4:]: v for k, v in synth.state_dict(42, 64, 50, 30).items() if k.startswith('cnn.')}
    img = synth.images(5, 4, 224).cuda()
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(M, 'EVAL_FUSE_POOL', fuse)
        m = M.CNN(2, 'max', precision='bf16')
        m.load_state_dict(sd)
        m = m.cuda().eval()
        with torch.no_grad():
            outs.append(m(img))
    assert torch.equal(outs[0], outs[1])
