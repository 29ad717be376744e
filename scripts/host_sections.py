"""Host (CPU) time of each section of one training step: tiny batch, so the GPU never back-pressures the issue loop;
perf_counter around the autograd Functions' forward / backward and the optimizer tail.  python scripts/host_sections.py"""
import collections, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqa_model
from vqa_amd import synth, cnn as CN, gru as GR, vqa_model as VM, optim as OP
from vqa_amd.optim import Trainer

acc = collections.defaultdict(float)


def timed(cls, meth, name):
    fn = getattr(cls, meth)
    f = fn.__func__ if hasattr(fn, '__func__') else fn

    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
    setattr(cls, meth, staticmethod(wrap))


for cls, nm in ((CN._CnnFunction, 'cnn'), (GR._GruFunction, 'gru'), (VM._HeadFunction, 'head')):
    timed(cls, 'forward', nm + '.fwd')
    timed(cls, 'backward', nm + '.bwd')
_opt = OP.FlatAdamax.step
def opt_step(self):
    t0 = time.perf_counter(); _opt(self); acc['adamax'] += time.perf_counter() - t0
OP.FlatAdamax.step = opt_step

with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
    torch.manual_seed(42)
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1, 'train_question_path': qpath},
                      lpath, 256, 2, 0.0, 'max', 'relu').cuda().train()
tr = Trainer(m)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = (synth.images(1, B, 224).cuda(), synth.questions(1, B, 14, 10000).cuda(), synth.targets(1, B, 3129).cuda())
with torch.cuda.stream(tr.stream):
    for _ in range(10):
        tr.step(*batch)
    torch.cuda.synchronize()
    acc.clear()
    N = 50
    t0 = time.perf_counter()
    for _ in range(N):
        tr.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
tot = (t1 - t0) / N * 1e3
print(f'B={B}: issue {tot:.3f} ms/step (done {(t2 - t0) / N * 1e3:.3f})')
named = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f'  {k:10s} {v / N * 1e3:.3f} ms')
    named += v / N * 1e3
print(f'  {"other":10s} {tot - named:.3f} ms   (autograd engine, Trainer glue, module forward)')
