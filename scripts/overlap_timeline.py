"""Timeline of one training step WITH its side streams running (per-launch path, events on every launching stream
around each C-ABI call, all read against one reference event): where does the trainer's stream wait, and how much
slower do its kernels run under the side streams' kernels than alone?   python scripts/overlap_timeline.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqa_model
from vqa_amd import synth, _lib as L
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
rec = []
raw_call = L._call_raw


def call(name, *args, tag=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = L._raw_stream()
    e0.record()
    raw_call(name, *args)
    e1.record()
    rec.append((name, tag, s, e0, e1))


with torch.cuda.stream(tr.stream):
    for _ in range(15):
        tr.step(*batch)
    L._PROFILE = []                      # (plans off: the per-launch path)
    for _ in range(3):
        tr.step(*batch)                  # per-launch path warm
    torch.cuda.synchronize()
    L.call = call
    marks = []
    ref = torch.cuda.Event(enable_timing=True)
    ref.record()
    for i in range(3):
        marks.append(len(rec))
        tr.step(*batch)
    marks.append(len(rec))
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
print('3 steps:', ref.elapsed_time(end), 'ms')
main = tr.stream.cuda_stream
lo, hi = marks[1], marks[2]
rows = []
for name, tag, s, e0, e1 in rec[lo:hi]:
    t = tag() if callable(tag) else tag
    rows.append((ref.elapsed_time(e0), ref.elapsed_time(e1), s, name, (t or {}).get('kind', ''), (t or {}).get('shape', '')))
t0 = min(r[0] for r in rows)
streams = sorted({r[2] for r in rows})
names = {s: ('main' if s == main else f's{streams.index(s)}') for s in streams}
prev = {s: t0 for s in streams}
busy = {s: 0.0 for s in streams}
print(f'step: {max(r[1] for r in rows) - t0:.3f} ms, {len(rows)} calls')
for a, b, s, name, kind, shape in sorted(rows):
    gap = a - prev[s]
    busy[s] += b - a
    prev[s] = b
    print(f'{names[s]:5s} t={(a - t0) * 1e3:8.1f} dur={(b - a) * 1e3:7.1f} gap={gap * 1e3:7.1f} {name[4:]:28s} {kind:8s} {shape}')
for s in streams:
    print(names[s], 'busy', round(busy[s], 3), 'ms')
