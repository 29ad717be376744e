"""Per-parameter errors of the HIP path against the B=32 golden (fp32 and bf16), and where the head's attention-layer
gradients lose accuracy in the whole model: encoder outputs vs golden, then the head alone on the GOLDEN encoder
outputs."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vqa_model
from oracle import vqa_oracle as O
from tests.helpers import build_model, load_case, rel_err, sample_idx

c = load_case('h1024_p2_max_b32'); g = c['golden']
img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
with tempfile.TemporaryDirectory() as tmp:
    for prec in ('fp32', 'bf16'):
        m = build_model(c, tmp, prec).cuda().train()
        v = m.cnn(img); h = m.gru(q)
        print(prec, 'cnn_out', rel_err(v.detach().cpu(), g['cnn_out']), 'gru_out', rel_err(h.detach().cpu(), g['gru_out']))
        m.zero_grad(set_to_none=True)
        lo = m(img, q); vqa_model.SoftScoreBCELoss()(lo, t).backward()
        for k, p in m.named_parameters():
            f = p.grad.reshape(-1).cpu().numpy(); idx = sample_idx(f.size)
            if float(g['gnorm/' + k]) < 1e-3: continue
            print(f'  {prec} {k:45s} {rel_err(f[idx], g["gsamp/"+k]):.2e} norm {abs(np.linalg.norm(f.astype(np.float64))-float(g["gnorm/"+k]))/float(g["gnorm/"+k]):.2e}')
        # head alone on the golden encoder outputs, against the oracle head on the same
        m.zero_grad(set_to_none=True)
        vg = torch.from_numpy(g['cnn_out']).cuda().requires_grad_(True); hg = torch.from_numpy(g['gru_out']).cuda().requires_grad_(True)
        vqa_model.SoftScoreBCELoss()(m.head(vg, hg), t).backward()
        p = {k: (x.clone().requires_grad_(True) if O.is_param(k) else x) for k, x in c['sd'].items() if not k.startswith(('cnn.', 'gru.'))}
        vr = torch.from_numpy(g['cnn_out']).requires_grad_(True); hr = torch.from_numpy(g['gru_out']).requires_grad_(True)
        O.bce_sum(O.head_forward(p, vr, hr), c['targets']).backward()
        print(f'  {prec} head on golden encoders: dv {rel_err(vg.grad.cpu(), vr.grad):.2e} dq {rel_err(hg.grad.cpu(), hr.grad):.2e}')
        for k, prm in m.named_parameters():
            if not k.startswith(('cnn.', 'gru.')):
                print(f'    {prec} {k:45s} {rel_err(prm.grad.cpu(), p[k].grad):.2e}')
        pp = m.head  # attention sharpness
    a = torch.softmax(torch.zeros(1), 0)
