"""One and several fp32 Trainer steps, dumping loss / gradient norms / running statistics: run under VQA_CONV_PAIR=0 and 1
and compare (python scripts/pair_step.py out.pt)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, vqa_model
from vqa_amd import synth
from vqa_amd.optim import Trainer

dev = torch.device('cuda', 0)
prec = os.environ.get('PREC', 'fp32')
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
    torch.manual_seed(42)
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1,
                       'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu', precision=prec)
m = m.to(dev).train()
tr = Trainer(m)
B = 128
batch = (synth.images(100, B, 224).to(dev), synth.questions(100, B, 14, 10000).to(dev), synth.targets(100, B, 3129).to(dev))
out = {}
for step in range(3):
    loss = tr.step(*batch, flip=False)
    torch.cuda.synchronize()
    out[f'loss{step}'] = loss.detach().cpu()
    out[f'grad{step}'] = tr.flat.grad.detach().cpu().clone()
    for k, b in m.named_buffers():
        if 'num_batches' not in k:
            out[f'{k}.{step}'] = b.detach().cpu().clone()
torch.save(out, sys.argv[1])
