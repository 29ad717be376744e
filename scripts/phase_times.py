"""GPU-side phase durations of the training step (events on the trainer's stream, no profiler):
CNN forward | head forward + loss + head backward | CNN backward | tail (join, clip + Adamax), against host issue time.
python scripts/phase_times.py [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqa_model
from vqa_amd import synth, cnn as CN
from vqa_amd.optim import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))

fwd0 = CN._CnnFunction.forward
bwd0 = CN._CnnFunction.backward
class Wrapped(CN._CnnFunction):
    @staticmethod
    def forward(ctx, *a):
        mark('cnn_fwd_start'); out = fwd0(ctx, *a); mark('cnn_fwd_end'); return out
    @staticmethod
    def backward(ctx, *g):
        mark('cnn_bwd_start'); out = bwd0(ctx, *g); mark('cnn_bwd_end'); return out
CN._CnnFunction = Wrapped
with torch.cuda.stream(tr.stream):
    for _ in range(10):
        tr.step(*batch)
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        mark('step_start')
        tr.step(*batch)
    mark('end')
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'{steps} steps: issued in {(t1 - t0) / steps * 1e3:.3f} ms/step, done after {(t2 - t0) / steps * 1e3:.3f} ms/step')
import collections
gpu = collections.defaultdict(float); host = collections.defaultdict(float)
for (n0, e0, h0), (n1, e1, h1) in zip(marks[:-1], marks[1:]):
    gpu[n0 + ' -> ' + n1] += e0.elapsed_time(e1); host[n0 + ' -> ' + n1] += (h1 - h0) * 1e3
for k in gpu:
    print(f'{k:34s} gpu {gpu[k] / steps:7.3f} ms   host {host[k] / steps:7.3f} ms')
# how far the host runs ahead of the GPU at each mark of the last step: host time of the mark vs GPU completion
