"""How far is the HIP bf16 step's gradient from the ORACLE run with the same bf16 rounding points (fp32 arithmetic on
the CPU), tensor by tensor, compared with how far either is from the fp32 golden?  B = 32, H = 1024 golden case.
python scripts/bf16_direct.py"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_train_parity as T
import vqa_model
from oracle import vqa_oracle as O

c = T.load_case('h1024_p2_max_b32')
g = c['golden']
m = T.build_model(c, pathlib.Path(tempfile.mkdtemp()), 'bf16').cuda().train()
logits = m(c['images'].cuda(), c['questions'].cuda())
vqa_model.SoftScoreBCELoss()(logits, c['targets'].cuda()).backward()
p = {k: (v.detach().clone().requires_grad_(True) if O.is_param(k) else v) for k, v in c['sd'].items()}
v = O.cnn_forward(p, c['images'].float(), c['padding'], c['pooling'], True, None, None, act_dtype=torch.bfloat16)
lo = O.head_forward(p, v, O.gru_forward(p, c['questions']), None, None)
names = [k for k in p if O.is_param(k)]
grads = dict(zip(names, torch.autograd.grad(O.bce_sum(lo, c['targets']), [p[k] for k in names])))
print(f'{"tensor":52s} hip-vs-bf16oracle  hip-vs-fp32golden  bf16oracle-vs-fp32golden')
for k, prm in m.named_parameters():
    if T._is_conv_bias(k) or k.endswith('weight_g'):
        continue
    hip = prm.grad.detach().reshape(-1).cpu()
    orc = grads[k].reshape(-1)
    idx = T.sample_idx(hip.numel())
    e_direct = T.rel_err(hip, orc)
    e_hip = T.rel_err(hip.numpy()[idx], g[f'gsamp/{k}'])
    e_orc = T.rel_err(orc.numpy()[idx], g[f'gsamp/{k}'])
    print(f'{k:52s} {e_direct:10.3e} {e_hip:18.3e} {e_orc:18.3e}')
