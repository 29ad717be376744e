"""Run the same fp32 training backward twice on one module and list gradients that are not bit-identical."""
import sys, tempfile, pathlib, torch
sys.path.insert(0, '.')
from tests.helpers import build_model, load_case
import vqa_model
c = load_case('h64_p5_max')
m = build_model(c, pathlib.Path(tempfile.mkdtemp()), sys.argv[1] if len(sys.argv) > 1 else 'fp32').cuda().train()
img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
runs = []
for it in range(3):
    m.zero_grad(set_to_none=True)
    vqa_model.SoftScoreBCELoss()(m(img, q), t).backward()
    torch.cuda.synchronize()
    runs.append({k: p.grad.clone() for k, p in m.named_parameters()})
for k in runs[0]:
    d = max(float((runs[i][k] - runs[0][k]).abs().max()) for i in (1, 2))
    if d > 0: print(k, d)
print('done')
