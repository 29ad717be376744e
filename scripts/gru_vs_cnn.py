"""Does the head wait for the question encoder?  Events at the end of the GRU forward (side stream) and of the CNN
forward (trainer's stream), and likewise for the two backwards; positive = the GRU side finished LATER.
python scripts/gru_vs_cnn.py [steps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqa_model
from vqa_amd import synth, cnn as CN, gru as GR
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
ev = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()                      # on the CURRENT stream: the side stream inside the GRU, the trainer's in the CNN
    ev.setdefault(name, []).append(e)

def wrap(fn_cls, tag):
    f0, b0 = fn_cls.forward, fn_cls.backward
    class W(fn_cls):
        @staticmethod
        def forward(ctx, *a):
            mark(tag + '_fwd_start'); out = f0(ctx, *a); mark(tag + '_fwd_end'); return out
        @staticmethod
        def backward(ctx, *g):
            mark(tag + '_bwd_start'); out = b0(ctx, *g); mark(tag + '_bwd_end'); return out
    return W
CN._CnnFunction = wrap(CN._CnnFunction, 'cnn')
gname = [n for n in dir(GR) if n.startswith('_') and n.endswith('Function')]
for n in gname:
    setattr(GR, n, wrap(getattr(GR, n), n.strip('_').lower()))
with torch.cuda.stream(tr.stream):
    for _ in range(10):
        tr.step(*batch)
    torch.cuda.synchronize()
    ev.clear()
    for _ in range(steps):
        tr.step(*batch)
    torch.cuda.synchronize()
print('functions wrapped in gru.py:', gname)
def avg(a, b):
    return sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b])) / len(ev[a])
keys = sorted(ev)
print(keys)
g = [k[:-10] for k in keys if k.endswith('_fwd_start') and not k.startswith('cnn')]
for t in g:
    print(f'{t}: forward ends {avg("cnn_fwd_end", t + "_fwd_end"):+.3f} ms after the CNN forward ends '
          f'(CNN forward {avg("cnn_fwd_start", "cnn_fwd_end"):.3f} ms, {t} forward {avg(t + "_fwd_start", t + "_fwd_end"):.3f} ms)')
    if t + '_bwd_end' in ev:
        print(f'{t}: backward ends {avg("cnn_bwd_end", t + "_bwd_end"):+.3f} ms after the CNN backward ends '
              f'(CNN backward {avg("cnn_bwd_start", "cnn_bwd_end"):.3f} ms, {t} backward {avg(t + "_bwd_start", t + "_bwd_end"):.3f} ms; '
              f'{t} backward starts {avg("cnn_bwd_start", t + "_bwd_start"):+.3f} ms after the CNN backward starts)')
