"""Experiment: how much does replaying the whole training step as one HIP graph gain over eager launches?"""
import sys, time, tempfile, torch
sys.path.insert(0, '.')
from bench import make_fixture
from vqa_amd import synth
from vqa_amd.optim import Trainer
import vqa_model
dev = torch.device('cuda', 0)
classes, vocab, B = 3129, 10000, 256
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = make_fixture(tmp, vocab, classes)
    torch.manual_seed(42)
    model = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1, 'train_question_path': qpath},
                          lpath, 256, 2, 0.0, 'max', 'relu', precision='bf16')
model = model.to(dev).train()
tr = Trainer(model)
batch = (synth.images(1234, B, 224).to(dev), synth.questions(1234, B, 14, vocab).to(dev), synth.targets(1234, B, classes).to(dev))
for _ in range(5): tr.step(*batch, flip=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): loss = tr.step(*batch, flip=False)
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t0) / 20 * 1e3, float(loss))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): tr.step(*batch, flip=False)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    loss = tr.step(*batch, flip=False)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print('graph ms/step', (time.perf_counter() - t0) / 20 * 1e3, float(loss))
