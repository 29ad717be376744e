"""fp32 accuracy of the HIP encoder against an fp64 run of the oracle, beside the reference's own fp32 (golden)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vqa_oracle as O
from tests.helpers import build_model, load_case, rel_err

c = load_case('h1024_p2_max_b32'); g = c['golden']
dt = torch.float64
p = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in c['sd'].items()}
inter = {}
v64 = O.cnn_forward(p, c['images'].to(dt), c['padding'], c['pooling'], True, None, inter)
h64 = O.gru_forward(p, c['questions'])
p32 = c['sd']
inter32 = {}
v32 = O.cnn_forward(p32, c['images'].float(), c['padding'], c['pooling'], True, None, inter32)
print('inter keys', list(inter)[:12])
with tempfile.TemporaryDirectory() as tmp:
    m = build_model(c, tmp, 'fp32').cuda().train()
    v = m.cnn(c['images'].cuda()); h = m.gru(c['questions'].cuda())
print('cnn_out vs fp64:  hip', rel_err(v.detach().cpu(), v64), ' golden', rel_err(g['cnn_out'], v64), ' oracle32', rel_err(v32, v64))
print('gru_out vs fp64:  hip', rel_err(h.detach().cpu(), h64), ' golden', rel_err(g['gru_out'], h64))
for k in inter:
    if k in inter32 and torch.is_tensor(inter[k]):
        print(f'  oracle32 vs fp64 {k:20s} {rel_err(inter32[k], inter[k]):.2e}')

# ReLU decisions of the attention layer (vqa_model.py:88) that the two fp32 encoders' outputs flip against the fp64 run
W = O.weight_norm_w(p['linear_inside_relu_attention.weight_g'], p['linear_inside_relu_attention.weight_v'])
b = p['linear_inside_relu_attention.bias']
def pre(vv, hh):
    vv, hh = torch.as_tensor(np.asarray(vv)).to(dt), torch.as_tensor(np.asarray(hh)).to(dt)
    return vv @ W[:, :256].T + (hh @ W[:, 256:].T + b)[:, None, :]
ref = pre(v64, h64)
for name, vv, hh in (('hip', v.detach().cpu(), h.detach().cpu()), ('golden', g['cnn_out'], g['gru_out'])):
    x = pre(vv, hh)
    flips = (x > 0) != (ref > 0)
    print(f'{name}: {int(flips.sum())} of {flips.numel()} attention units on the other side of zero; |pre| of those:',
          [f'{float(t):.1e}' for t in ref[flips].abs()[:8]], ' typical |pre|', float(ref.abs().median()))

# structure of the two encoders' errors
eh = (v.detach().cpu().to(dt) - v64); er = (torch.as_tensor(g['cnn_out']).to(dt) - v64)
print('corr(e_hip, e_ref)', float((eh * er).sum() / (eh.norm() * er.norm())))
for name, e in (('hip', eh), ('golden', er)):
    pc = e.mean((0, 1))                 # per-channel mean error
    print(name, 'norm', float(e.norm() / v64.norm()), ' per-channel-mean part', float((pc.norm() * (e.shape[0] * e.shape[1]) ** 0.5) / e.norm()),
          ' max abs', float(e.abs().max()), ' per-sample norms', [f'{float(x):.1e}' for x in (e.reshape(32, -1).norm(dim=1) / v64.reshape(32, -1).norm(dim=1))[:8]])
    # sensitivity: d(attention scores) for this error
    s_err = (pre(v64 + e, h64).clamp_min(0) - ref.clamp_min(0)) @ O.weight_norm_w(p['linear_after_relu_attention.weight_g'], p['linear_after_relu_attention.weight_v']).T
    s_ref = ref.clamp_min(0) @ O.weight_norm_w(p['linear_after_relu_attention.weight_g'], p['linear_after_relu_attention.weight_v']).T
    print('   score error: max', float(s_err.abs().max()), ' spread over regions (what the softmax sees)', float((s_err - s_err.mean(1, keepdim=True)).abs().max()), ' score scale', float(s_ref.abs().max()))

# sensitivity of the head's gradients to each encoder's output error, all in fp64 arithmetic
def head_grads(vv, hh):
    ph = {k: (x.to(dt).detach().clone().requires_grad_(O.is_param(k)) if x.is_floating_point() else x)
          for k, x in c['sd'].items() if not k.startswith(('cnn.', 'gru.'))}
    loss = O.bce_sum(O.head_forward(ph, vv, hh), c['targets'].to(dt))
    names = [k for k in ph if O.is_param(k)]
    return dict(zip(names, torch.autograd.grad(loss, [ph[k] for k in names])))
base = head_grads(v64, h64)
for name, vv, hh in (('hip', v.detach().cpu().to(dt), h.detach().cpu().to(dt)),
                     ('golden', torch.as_tensor(g['cnn_out']).to(dt), torch.as_tensor(g['gru_out']).to(dt)),
                     ('hip v, fp64 q', v.detach().cpu().to(dt), h64), ('fp64 v, hip q', v64, h.detach().cpu().to(dt))):
    gg = head_grads(vv, hh)
    print(name, {k.replace('linear_', '')[:28]: f'{rel_err(gg[k], base[k]):.1e}' for k in gg if 'attention' in k or k == 'fc.weight_v'})
