import ctypes, torch
from vqa_amd import cnn as M, synth, _lib as L
from oracle import vqa_oracle as O
sd = synth.state_dict(42, 64, 50, 30)
img = synth.images(7, 2, 224)
inter = {}
O.cnn_forward(sd, img.float(), 2, 'max', True, None, inter, act_dtype=torch.bfloat16)
dtype = torch.bfloat16
x = M.to_nhwc(img.cuda(), dtype)
rel = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
for li, ci in enumerate(O.CONV_IDX):
    w = sd[f'cnn.convolutions.{ci}.weight'].cuda(); b = sd[f'cnn.convolutions.{ci}.bias'].cuda()
    g = sd[f'cnn.convolutions.{ci+1}.weight'].cuda(); be = sd[f'cnn.convolutions.{ci+1}.bias'].cuda()
    rm = sd[f'cnn.convolutions.{ci+1}.running_mean'].cuda().clone(); rv = sd[f'cnn.convolutions.{ci+1}.running_var'].cuda().clone()
    pad = 2 if li in (0, 2, 4, 6) else 0
    pool = 1 if li % 2 == 1 else 0
    cout = M.CHANNELS[li + 1]
    wpk = M.pack_weights(w, x.shape[3], dtype, False)
    y, stats = M.conv_fwd(x, wpk, b, cout, pad, True)
    f = lambda: torch.empty(cout, device='cuda')
    mean, invstd, scale, shift = f(), f(), f(), f()
    cnt = float(y.shape[0] * y.shape[1] * y.shape[2])
    L.lib().vqa_bn_finalize(L.stream(), L.ptr(stats), stats.shape[0], cout, ctypes.c_double(cnt), L.ptr(g), L.ptr(be), L.ptr(rm), L.ptr(rv), ctypes.c_float(0.1), ctypes.c_float(1e-5), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(shift), None)
    a = M.bn_act_pool(y, scale, shift, pool)
    ref = inter[f'cnn.act{li}']
    mine = a.float().cpu().permute(0, 3, 1, 2)
    d = (mine - ref).abs()
    print(li, 'act err', '%.2e' % rel(mine, ref), 'frac differing', '%.4f' % float((d > 0).float().mean()), 'max', float(d.max()), 'mean|ref|', float(ref.abs().mean()))
    x = a
