"""How long does the CPU need to ISSUE one training step (no sync inside), against the GPU's time per step?
python scripts/cpu_issue_time.py   (on the GPU box)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqa_model
from vqa_amd import synth
from vqa_amd.optim import Trainer

dev = torch.device('cuda', 0)
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
    torch.manual_seed(42)
    model = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1,
                           'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu')
model = model.to(dev).train()
tr = Trainer(model)
batch = (synth.images(1234, 256, 224).to(dev), synth.questions(1234, 256, 14, 10000).to(dev),
         synth.targets(1234, 256, 3129).to(dev))
for _ in range(10):
    tr.step(*batch)
torch.cuda.synchronize()
for n in (3, 6):
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{n} steps: issued in {(t1 - t0) / n * 1e3:.2f} ms/step, done after {(t2 - t0) / n * 1e3:.2f} ms/step')
